/* TEST INFRASTRUCTURE ONLY (oracle/).  Scalar, single-threaded CPU restatement of the reference Voxelyze
 * time-stepper in plain C.  It is the checker for the HIP engine, never the thing measured or shipped:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
 *
 * PARITY PINNED against the reference itself: tests/test_oracle_vs_reference.py compares this code,
 * voxel by voxel, with traces written by the unmodified reference C++ (oracle/_ref/vxprobe, compiled from
 * /root/reference by oracle/Makefile) and with the result XMLs of the reference binary (tests/golden/).
 * Operation order and association follow the reference expressions exactly (no FMA: build with
 * -ffp-contract=off) so that, with the same libm, the state matches the reference bit for bit.
 *
 * Paths below abbreviate: VX/ = evosoro/_voxcad/Voxelyze/, LW/ = evosoro/_voxcad_land_water/Voxelyze/.
 * Scope: what evosoro's writer can switch on (App. B of SURVEY.md).  Not restated (all off by default and
 * never written by evosoro/tools/read_write_voxelyze.py): boundary-condition regions, volume effects,
 * plasticity/failure, blending, max-velocity limit, equilibrium mode, environmental sources, needle-in-haystack,
 * limited floor, and of the _voxcad development model the velocity-adjusted part (NumTimeStepsInWindow > 0).  The
 * growth / development tags themselves (InitialVoxelSize, FinalVoxelSize, GrowthTime, StartGrowthTime, FinalPhaseOffset,
 * FinalTempAmpDamp, MidLifeFreezeTime, MinGrowthTime) ARE restated (SCALE section below).
 */
#include "vx_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double x, y, z; } v3;
typedef struct { double w, x, y, z; } qt;

/* ---- Vec3D<double> / CQuat<double> helpers, VX/Utils/Vec3D.h ------------------------------------------- */
static v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }          /* :88 */
static v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }          /* :89 */
static v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }                               /* :90 */
static v3 vmul(v3 a, double f) { return V(f * a.x, f * a.y, f * a.z); }            /* :91 */
static v3 vdiv(v3 a, double f) { double inv = 1.0 / f; return V(inv * a.x, inv * a.y, inv * a.z); } /* :93 */
static double vlen2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }            /* :131 */
static double vlen(v3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }       /* :130 */

static qt Q(double w, double x, double y, double z) { qt r = {w, x, y, z}; return r; }
static qt qconj(qt a) { return Q(a.w, -a.x, -a.y, -a.z); }                         /* :249 */
static qt qmul(qt a, qt f)                                                         /* :193 */
{
    return Q(a.w * f.w - a.x * f.x - a.y * f.y - a.z * f.z,
             a.w * f.x + a.x * f.w + a.y * f.z - a.z * f.y,
             a.w * f.y - a.x * f.z + a.y * f.w + a.z * f.x,
             a.w * f.z + a.x * f.y - a.y * f.x + a.z * f.w);
}
static v3 qrotinv(qt q, v3 f)                                                      /* RotateVec3DInv :300-314 */
{
    double tw = q.x * f.x + q.y * f.y + q.z * f.z;
    double tx = q.w * f.x - q.y * f.z + q.z * f.y;
    double ty = q.w * f.y + q.x * f.z - q.z * f.x;
    double tz = q.w * f.z - q.x * f.y + q.y * f.x;
    return V(tw * q.x + tx * q.w + ty * q.z - tz * q.y,
             tw * q.y - tx * q.z + ty * q.w + tz * q.x,
             tw * q.z + tx * q.y - ty * q.x + tz * q.w);
}

#define VX_PI 3.14159265358979                 /* Vec3D.h:21 */
#define HYST 1.1                               /* VEC3D_HYSTERESIS_FACTOR :22 */
#define DISCARD_ANGLE_RAD 1e-7                 /* :47 */
#define SMALL_ANGLE_RAD 1.732e-2               /* :48 */
#define W_THRESH_ACOS2SQRT 0.9988              /* :49 */
#define SA_BOND_BEND_RAD 0.05                  /* VXS_BondInternal.h:23 */
#define SA_BOND_EXT_PERC 1.30                  /* VXS_BondInternal.h:26 */
static double SMALL_ANGLE_W, SMALLISH_ANGLE_W, SLTHRESH_ACOS2SQRT; /* Vec3D.h:55-59, set in init_consts */
static void init_consts(void)
{
    SMALL_ANGLE_W = cos(SMALL_ANGLE_RAD * 0.5);
    SMALLISH_ANGLE_W = cos(HYST * SMALL_ANGLE_RAD * 0.5);
    SLTHRESH_ACOS2SQRT = 1.0 - W_THRESH_ACOS2SQRT * W_THRESH_ACOS2SQRT;
}

static qt q_from_angle_to_pos_x(v3 from)                                           /* :208-237 */
{
    qt q = Q(1, 0, 0, 0);
    if (from.x == 0 && from.y == 0 && from.z == 0) return q;
    double YoverX = from.y / from.x, ZoverX = from.z / from.x;
    if (YoverX < SMALL_ANGLE_RAD && YoverX > -SMALL_ANGLE_RAD && ZoverX < SMALL_ANGLE_RAD && ZoverX > -SMALL_ANGLE_RAD) {
        q.x = 0; q.y = 0.5 * ZoverX; q.z = -0.5 * YoverX;
        q.w = 1 + 0.5 * (-q.y * q.y - q.z * q.z);
        return q;
    }
    v3 n = from;
    double l = sqrt(n.x * n.x + n.y * n.y + n.z * n.z);                            /* NormalizeFast :117 */
    if (l > 0) { double li = 1.0 / l; n.x *= li; n.y *= li; n.z *= li; }
#ifdef VXO_HALF_ANGLE
    /* NOT the reference's evaluation: the same rotation written with the half-angle identities the HIP engine uses (kernels.hpp
     * from_angle_to_pos_x; DESIGN.md "Numerics").  Built only into libvxoracle_ha.so, with which tests/test_gpu_ledger.py measures
     * how far THIS ONE rewrite moves the reference algorithm's own answer over a whole run. */
    if (n.x < -0.999999999999995) return Q(0, 0, 1, 0);
    {
        double c = sqrt(0.5 + 0.5 * n.x), h = 0.5 / c;
        return Q(c, 0, n.z * h, -n.y * h);
    }
#endif
    double theta = acos(n.x);
    if (theta > VX_PI - DISCARD_ANGLE_RAD) return Q(0, 0, 1, 0);
    double AxisMagInv = 1.0 / sqrt(n.z * n.z + n.y * n.y);
    double a = 0.5 * theta, s = sin(a);
    q.w = cos(a); q.x = 0; q.y = n.z * AxisMagInv * s; q.z = -n.y * AxisMagInv * s;
    return q;
}
static v3 q_to_rotvec(qt q)                                                        /* ToRotationVector :270-285 */
{
    double sl = 1.0 - q.w * q.w;
    if (sl <= 0) return V(0, 0, 0);
    double wc = q.w > 1 ? 1 : q.w;
    if (sl < SLTHRESH_ACOS2SQRT) return vmul(vmul(V(q.x, q.y, q.z), 2.0), sqrt((2 - 2 * wc) / sl));
    return vdiv(vmul(vmul(V(q.x, q.y, q.z), 2.0), acos(wc)), sqrt(sl));
}

/* ---- simulation state ------------------------------------------------------------------------------------ */
typedef struct {
    /* constants, CVX_Voxel::SetMaterial VX/VX_Voxel.cpp:94-128 */
    int mat; v3 nom_pos; double nom_size;
    double mass, inertia, first_moment, E, nu, cte, mass_inv, inertia_inv, c_lin /*_2xSqMxExS*/, c_ang /*_2xSqIxExSxSxS*/;
    double mat_E;      /* Elastic_Mod of the material (stress model), distinct from E when <Stiffness> is evolved */
    double u_static, u_dynamic;
    int bond[6];       /* InternalBondIndices by BondDir PX,NX,PY,NY,PZ,NZ (VX_Enums.h:107-114), -1 none */
    /* float-typed per-voxel parameters, VX/VXS_Voxel.h:92-111 */
    float phase_offset, temp_amp_damp, temp_amplitude, temp_period, initial_voxel_size, start_growth_time, growth_time;
    float final_phase_offset, final_temp_amp_damp, final_voxel_size, onset_bound, termination_bound;
    /* state, VX/VXS_Voxel.h:126-145 */
    v3 pos, lin_mom, ang_mom, vel, ang_vel; qt angle; double scale, last_scale; int static_fric;
    v3 strain_pos, strain_neg;   /* StrainPosDirsCur / StrainNegDirsCur */
    v3 drag;                     /* LW DragForce */
    v3 corner_pos, corner_neg;   /* CornerPosCur / CornerNegCur (only the LW drag mesh reads them) */
    /* collision links in creation order: index into col[] */
    int ncol, capcol; int* col;
    /* nearby voxels (CalcNearby) as a sorted list for the exclusion test */
    int nnear; int* near;
} voxel;

typedef struct {
    int v1, v2, axis;  /* axis 1=X 2=Y 3=Z (VX_Enums.h Axis) */
    int homogeneous;
    double L, a1, a2, b1y, b2y, b3y, b1z, b2z, b3z;
    double sq_a1m1, sq_a1m2, sq_a2i1, sq_a2i2, sq_b1ym1, sq_b1ym2, sq_b1zm1, sq_b1zm2;
    double sq_b2yfm1, sq_b2yfm2, sq_b2zfm1, sq_b2zfm2, sq_b3yi1, sq_b3yi2, sq_b3zi1, sq_b3zi2;
    double cs_area1, cs_area2;
    /* state */
    int small_angle;
    v3 pos2, angle1, angle2, last_pos2, last_angle1, last_angle2;
    double strain_tot, strain_v1, strain_v2, stress;
    v3 f1, f2, m1, m2;
} ibond;

typedef struct { int v1, v2; double a1; v3 f1, f2; } cbond;
struct mvert { int nc; int vox[8]; int corner[8]; v3 v0, cur; };   /* lattice corner touched by 1..7 voxels */
struct mfacet { int vi[3]; int owner; v3 n; };

struct vxo_sim {
    vxo_model m;
    int nvox, nbond, nsurf, ncol, capcol;
    voxel* vox; ibond* bond; int* surf; cbond* col;
    double lat, opt_dt, dt, cur_time; int steps, status, cm_init;
    double max_disp_since_update; int col_enable_changed, rebuilds;
    double max_vox_vel; v3 cur_cm, ini_cm; double end_of_life_posterior_y;
    /* LW fluid drag mesh (LW/VX_MeshUtil.cpp:110-276) */
    int nmv, nmf; struct mvert* mv; struct mfacet* mf;
    /* SS.CMTraceTime / SS.CMTrace (VX_Sim.cpp:1537-1547) */
    int ntrace, captrace; double* trace;
};

static double bond_E(double E1, double E2) { return (E1 * E2 / (E1 + E2)) * 2; }    /* VX/VX_Bond.cpp:87 */

/* CVX_Voxel::SetMaterial VX/VX_Voxel.cpp:94-128 (Vox_E may be overridden afterwards by SetEMod) */
static void voxel_set_material(const vxo_model* m, voxel* v, int mat, double size)
{
    v->mat = mat; v->nom_size = size;
    double Volume = size * size * size;
    v->mass = Volume * m->mat_rho[mat];
    v->inertia = v->mass * (size * size) / 6;
    v->first_moment = v->mass * size / 2;
    v->E = m->mat_E[mat]; v->mat_E = m->mat_E[mat]; v->nu = m->mat_nu[mat]; v->cte = m->mat_cte[mat];
    v->u_static = m->mat_us[mat]; v->u_dynamic = m->mat_ud[mat];
    v->mass_inv = 1 / v->mass; v->inertia_inv = 1 / v->inertia;
    v->c_lin = 2 * sqrt(v->mass * v->E * size);
    v->c_ang = 2 * sqrt(v->inertia * v->E * size * size * size);
}

/* CVX_Bond::LinkVoxels + UpdateConstants, VX/VX_Bond.cpp:65-173 (LW/VX_Bond.cpp:75 for the homogeneity rule) */
static void bond_link(const vxo_sim* s, ibond* b, int i1, int i2, int axis)
{
    const voxel* p1 = &s->vox[i1]; const voxel* p2 = &s->vox[i2];
    memset(b, 0, sizeof(*b));
    b->v1 = i1; b->v2 = i2; b->axis = axis;
    b->homogeneous = (p1->mat == p2->mat);
    if (s->m.variant == 1) b->homogeneous = b->homogeneous && (p1->E == p2->E);
    double E1 = p1->E, E2 = p2->E, u1 = p1->nu, u2 = p2->nu;
    double E = bond_E(E1, E2), u;
    if (u1 == 0 && u2 == 0) u = 0; else u = (u1 * u2 / (u1 + u2)) * 2;
    double NominalSize = (p1->nom_size + p2->nom_size) * 0.5;
    double Lx = NominalSize, Ly = NominalSize, Lz = NominalSize;
    b->L = Lx;
    double G = E / (2 * (1 + u));
    double A = Ly * Lz;
    double Iy = Lz * Ly * Ly * Ly / 12;
    double Iz = Ly * Lz * Lz * Lz / 12;
    double J = Ly * Lz * (Ly * Ly + Lz * Lz) / 12;
    b->a1 = E * A / Lx; b->a2 = G * J / Lx;
    b->b1y = 12 * E * Iy / (Lx * Lx * Lx); b->b1z = 12 * E * Iz / (Lx * Lx * Lx);
    b->b2y = 6 * E * Iy / (Lx * Lx);       b->b2z = 6 * E * Iz / (Lx * Lx);
    b->b3y = 2 * E * Iy / Lx;              b->b3z = 2 * E * Iz / Lx;
    double M1 = p1->mass, M2 = p2->mass, FM1 = p1->first_moment, FM2 = p2->first_moment, I1 = p1->inertia, I2 = p2->inertia;
    b->sq_a1m1 = 2.0 * sqrt(b->a1 * M1);   b->sq_a1m2 = 2.0 * sqrt(b->a1 * M2);
    b->sq_a2i1 = 2.0 * sqrt(b->a2 * I1);   b->sq_a2i2 = 2.0 * sqrt(b->a2 * I2);
    b->sq_b1ym1 = 2.0 * sqrt(b->b1y * M1); b->sq_b1ym2 = 2.0 * sqrt(b->b1y * M2);
    b->sq_b1zm1 = 2.0 * sqrt(b->b1z * M1); b->sq_b1zm2 = 2.0 * sqrt(b->b1z * M2);
    b->sq_b2yfm1 = 2.0 * sqrt(b->b2y * FM1); b->sq_b2yfm2 = 2.0 * sqrt(b->b2y * FM2);
    b->sq_b2zfm1 = 2.0 * sqrt(b->b2z * FM1); b->sq_b2zfm2 = 2.0 * sqrt(b->b2z * FM2);
    b->sq_b3yi1 = 2.0 * sqrt(b->b3y * I1); b->sq_b3yi2 = 2.0 * sqrt(b->b3y * I2);
    b->sq_b3zi1 = 2.0 * sqrt(b->b3z * I1); b->sq_b3zi2 = 2.0 * sqrt(b->b3z * I2);
    /* CVXS_Bond::ResetBond VX/VXS_Bond.cpp:60-88, CVXS_BondInternal::ResetBond VX/VXS_BondInternal.cpp:40-53 */
    b->cs_area1 = b->cs_area2 = Ly * Lz;
    b->small_angle = 1;
}

/* ToXDirBond / ToOrigDirBond, VX/VX_Bond.h:45-48 */
static v3 to_xdir_v(int axis, v3 p) { if (axis == 2) return V(p.y, -p.x, p.z); if (axis == 3) return V(p.z, p.y, -p.x); return p; }
static qt to_xdir_q(int axis, qt q) { if (axis == 2) return Q(q.w, q.y, -q.x, q.z); if (axis == 3) return Q(q.w, q.z, q.y, -q.x); return q; }
static v3 to_orig_v(int axis, v3 p) { if (axis == 2) return V(-p.y, p.x, p.z); if (axis == 3) return V(-p.z, p.y, p.x); return p; }

/* CVXC_Material::GetModelStress for MDL_LINEAR, VX/VX_Object.cpp:1472-1482; LW passes Vox_E through a float
 * parameter (LW/VX_Object.cpp:1474, LW/VXS_Voxel.cpp CalcVoxMatStress) */
static double mat_stress(const vxo_sim* s, const voxel* v, double strain)
{
    if (s->m.variant == 1) { float e = (float)v->E; float t = (e > 0) ? e : (float)v->mat_E; return t * strain; }
    return v->mat_E * strain;
}

/* CVXS_BondInternal::UpdateBondStrain VX/VXS_BondInternal.cpp:189-307 (plasticity and volume effects off) */
static void bond_update_strain(vxo_sim* s, ibond* b, double strain)
{
    voxel* p1 = &s->vox[b->v1]; voxel* p2 = &s->vox[b->v2];
    b->strain_tot = strain;
    if (b->homogeneous) {
        b->stress = mat_stress(s, p1, strain);
        b->strain_v1 = b->strain_v2 = strain;
    } else {
        b->strain_v1 = strain; b->strain_v2 = strain;
        double S1 = mat_stress(s, p1, b->strain_v1), S2 = mat_stress(s, p2, b->strain_v2);
        int count = 0;
        double diff = (S1 >= S2) ? S1 - S2 : S2 - S1;
        double sum = S1 + S2; if (sum < 0) sum = -sum;
        while (diff > sum * .0005 && count < 3) {
            b->strain_v1 = 2 * S2 / (S1 + S2) * b->strain_v1;
            b->strain_v2 = 2 * S1 / (S1 + S2) * b->strain_v2;
            S1 = mat_stress(s, p1, b->strain_v1); S2 = mat_stress(s, p2, b->strain_v2);
            diff = (S1 >= S2) ? S1 - S2 : S2 - S1;
            sum = S1 + S2; if (sum < 0) sum = -sum;
            count++;
        }
        b->stress = (S1 + S2) / 2;
    }
    switch (b->axis) {                                                             /* SetStrainDir :300-304 */
    case 1: p1->strain_pos.x = b->strain_v1; p2->strain_neg.x = b->strain_v2; break;
    case 2: p1->strain_pos.y = b->strain_v1; p2->strain_neg.y = b->strain_v2; break;
    case 3: p1->strain_pos.z = b->strain_v1; p2->strain_neg.z = b->strain_v2; break;
    }
}

/* CVXS_BondInternal::AddDampForces VX/VXS_BondInternal.cpp:310-346 */
static void bond_add_damp(vxo_sim* s, ibond* b)
{
    if (s->dt != 0) {
        double BondZ = 0.5 * s->m.bond_damping_z;
        double DtInv = 1.0 / s->dt;
#ifdef VXO_FOLD_DAMP   /* instrument: the engine's folding of 1 / dt and BondDampingZ / 2 into the 2 sqrt(k m) constants (DBondClass) */
        {
        v3 RelVel2 = vsub(b->pos2, b->last_pos2), W1 = vsub(b->angle1, b->last_angle1), W2 = vsub(b->angle2, b->last_angle2);
        {
            double zi = BondZ * DtInv, zh = 0.5 * zi;
            double dA1 = b->sq_a1m1 * zi, dB1 = b->sq_b1ym1 * zi, dF1 = b->sq_b2yfm1 * zi, dA2 = b->sq_a1m2 * zi, dB2 = b->sq_b1ym2 * zi, dF2 = b->sq_b2yfm2 * zi;
            double dT1 = b->sq_a2i1 * zh, dG1 = b->sq_b2yfm1 * zh, dH1 = b->sq_b3yi1 * zh, dT2 = b->sq_a2i2 * zh, dG2 = b->sq_b2yfm2 * zh, dH2 = b->sq_b3yi2 * zh;
            b->f1 = vadd(b->f1, V(dA1 * RelVel2.x, dB1 * RelVel2.y - dF1 * (W1.z + W2.z), dB1 * RelVel2.z + dF1 * (W1.y + W2.y)));
            if (!b->homogeneous) b->f2 = vadd(b->f2, V(-dA2 * RelVel2.x, -dB2 * RelVel2.y + dF2 * (W1.z + W2.z), -dB2 * RelVel2.z - dF2 * (W1.y + W2.y)));
            b->m1 = vadd(b->m1, V(-dT1 * (W2.x - W1.x), dG1 * RelVel2.z + dH1 * (2 * W1.y + W2.y), -dG1 * RelVel2.y + dH1 * (2 * W1.z + W2.z)));
            b->m2 = vadd(b->m2, V(dT2 * (W2.x - W1.x), dG2 * RelVel2.z + dH2 * (W1.y + 2 * W2.y), -dG2 * RelVel2.y + dH2 * (W1.z + 2 * W2.z)));
        }
        b->last_pos2 = b->pos2; b->last_angle1 = b->angle1; b->last_angle2 = b->angle2;
        return;
        }
#endif
        v3 RelVel2 = vmul(vsub(b->pos2, b->last_pos2), DtInv);
        v3 W1 = vmul(vsub(b->angle1, b->last_angle1), DtInv);
        v3 W2 = vmul(vsub(b->angle2, b->last_angle2), DtInv);
        b->f1 = vadd(b->f1, vmul(V(b->sq_a1m1 * RelVel2.x,
                                   b->sq_b1ym1 * RelVel2.y - b->sq_b2zfm1 * (W1.z + W2.z),
                                   b->sq_b1zm1 * RelVel2.z + b->sq_b2yfm1 * (W1.y + W2.y)), BondZ));
        if (!b->homogeneous)
            b->f2 = vadd(b->f2, vmul(V(-b->sq_a1m2 * RelVel2.x,
                                       -b->sq_b1ym2 * RelVel2.y + b->sq_b2zfm2 * (W1.z + W2.z),
                                       -b->sq_b1zm2 * RelVel2.z - b->sq_b2yfm2 * (W1.y + W2.y)), BondZ));
        b->m1 = vadd(b->m1, vmul(V(-b->sq_a2i1 * (W2.x - W1.x),
                                   b->sq_b2zfm1 * RelVel2.z + b->sq_b3yi1 * (2 * W1.y + W2.y),
                                   -b->sq_b2yfm1 * RelVel2.y + b->sq_b3zi1 * (2 * W1.z + W2.z)), 0.5 * BondZ));
        b->m2 = vadd(b->m2, vmul(V(b->sq_a2i2 * (W2.x - W1.x),
                                   b->sq_b2zfm2 * RelVel2.z + b->sq_b3yi2 * (W1.y + 2 * W2.y),
                                   -b->sq_b2yfm2 * RelVel2.y + b->sq_b3zi2 * (W1.z + 2 * W2.z)), 0.5 * BondZ));
    }
    b->last_pos2 = b->pos2; b->last_angle1 = b->angle1; b->last_angle2 = b->angle2;
}

/* CVXS_BondInternal::CalcLinForce VX/VXS_BondInternal.cpp:56-187 */
static void bond_update(vxo_sim* s, ibond* b)
{
    const voxel* p1 = &s->vox[b->v1]; const voxel* p2 = &s->vox[b->v2];
    v3 CurXRelPos = to_xdir_v(b->axis, vsub(p2->pos, p1->pos));
    qt CurXAng1 = to_xdir_q(b->axis, p1->angle), CurXAng2 = to_xdir_q(b->axis, p2->angle);
    v3 Rel = qrotinv(CurXAng1, CurXRelPos);                       /* Ang1AlignedRelPos */
    qt NewAng2 = qmul(qconj(CurXAng1), CurXAng2);
    double NomDistance = (p1->scale + p2->scale) * 0.5;

    int changed = 0;
    double SmallTurn = (fabs(Rel.z) + fabs(Rel.y)) / Rel.x;
    double ExtendPerc = Rel.x / NomDistance;
    if (!b->small_angle && NewAng2.w > SMALL_ANGLE_W && SmallTurn < SA_BOND_BEND_RAD && ExtendPerc < SA_BOND_EXT_PERC) { b->small_angle = 1; changed = 1; }
    else if (b->small_angle && (!(NewAng2.w > SMALLISH_ANGLE_W) || SmallTurn > HYST * SA_BOND_BEND_RAD || ExtendPerc > HYST * SA_BOND_EXT_PERC)) { b->small_angle = 0; changed = 1; }

    qt TotalRot;
    if (b->small_angle) {
        b->angle1 = V(0, 0, 0);
        b->angle2 = q_to_rotvec(NewAng2);
        Rel.x -= NomDistance;
        b->pos2 = Rel;
        TotalRot = qconj(CurXAng1);
    } else {
        qt Pos2AlignedRotAng = q_from_angle_to_pos_x(Rel);
        TotalRot = qmul(Pos2AlignedRotAng, qconj(CurXAng1));
        double Length = vlen(CurXRelPos);
        b->pos2 = V(Length - NomDistance, 0, 0);
        b->angle1 = q_to_rotvec(Pos2AlignedRotAng);
        b->angle2 = q_to_rotvec(qmul(TotalRot, CurXAng2));
    }
    bond_update_strain(s, b, b->pos2.x / b->L);

    b->f1 = V(b->stress * (b->cs_area1 + b->cs_area2) / 2,
              b->b1z * b->pos2.y - b->b2z * (b->angle1.z + b->angle2.z),
              b->b1y * b->pos2.z + b->b2y * (b->angle1.y + b->angle2.y));
    b->f2 = vneg(b->f1);
    b->m1 = V(b->a2 * (b->angle1.x - b->angle2.x),
              b->b2z * b->pos2.z + b->b3y * (2 * b->angle1.y + b->angle2.y),
              -b->b2y * b->pos2.y + b->b3z * (2 * b->angle1.z + b->angle2.z));
    b->m2 = V(b->a2 * (b->angle2.x - b->angle1.x),
              b->b2z * b->pos2.z + b->b3y * (b->angle1.y + 2 * b->angle2.y),
              -b->b2y * b->pos2.y + b->b3z * (b->angle1.z + 2 * b->angle2.z));
    /* StrainEnergy (CALCSTAT_ALL, VX/VX_Sim.cpp:97) feeds nothing on this path: omitted */
    if (!changed) bond_add_damp(s, b);

    b->f1 = qrotinv(TotalRot, b->f1);
    if (!b->homogeneous) b->f2 = qrotinv(TotalRot, b->f2);
    b->m1 = qrotinv(TotalRot, b->m1);
    b->m2 = qrotinv(TotalRot, b->m2);
    b->f1 = to_orig_v(b->axis, b->f1);
    if (b->homogeneous) b->f2 = vneg(b->f1); else b->f2 = to_orig_v(b->axis, b->f2);
    b->m1 = to_orig_v(b->axis, b->m1);
    b->m2 = to_orig_v(b->axis, b->m2);
}

/* CVXS_BondCollision::CalcContactForce VX/VXS_BondCollision.cpp:41-59 */
static void col_update(vxo_sim* s, cbond* c)
{
    const voxel* p1 = &s->vox[c->v1]; const voxel* p2 = &s->vox[c->v2];
    v3 Pos2 = vsub(p2->pos, p1->pos);
    double NomDist = (p1->scale + p2->scale) * 0.75;
    double RelDist = NomDist - vlen(Pos2);
    if (RelDist > 0) { c->f2 = vmul(vmul(vdiv(Pos2, vlen(Pos2)), c->a1), RelDist); c->f1 = vneg(c->f2); }
    else { c->f2 = V(0, 0, 0); c->f1 = V(0, 0, 0); }
}

static int is_nearby(const voxel* v, int other)
{
    int lo = 0, hi = v->nnear - 1;
    while (lo <= hi) { int mid = (lo + hi) / 2; if (v->near[mid] == other) return 1; if (v->near[mid] < other) lo = mid + 1; else hi = mid - 1; }
    return 0;
}
static void vox_link_col(voxel* v, int ci)
{
    if (v->ncol == v->capcol) { v->capcol = v->capcol ? 2 * v->capcol : 8; v->col = (int*)realloc(v->col, sizeof(int) * v->capcol); }
    v->col[v->ncol++] = ci;
}

/* CVX_Sim::CalcL1Bonds VX/VX_Sim.cpp:2357-2413 (the `CurColSystem == COL_SURFACE || COL_SURFACE_HORIZON`
 * test at :2369 is always true, so only surface voxels are ever paired) */
static void calc_l1_bonds(vxo_sim* s, double Dist)
{
    double FilterDist = Dist * 1.5 * s->lat, FilterDist2 = FilterDist * FilterDist;
    s->rebuilds++;
    for (int i = 0; i < s->nvox; i++) s->vox[i].ncol = 0;                           /* DeleteCollisionBonds :462-471 */
    s->ncol = 0;
    for (int i = 0; i < s->nsurf; i++) {
        int i1 = s->surf[i]; voxel* p1 = &s->vox[i1];
        for (int j = i + 1; j < s->nsurf; j++) {
            int i2 = s->surf[j]; voxel* p2 = &s->vox[i2];
            double Dist2 = vlen2(vsub(p1->pos, p2->pos));
            if (Dist2 < FilterDist2 && !is_nearby(p1, i2)) {
                double ActDist = Dist * (p1->scale + p1->scale) * 0.5;              /* pV1 twice, :2382 */
                if (Dist2 < ActDist * ActDist) {                                    /* CreateColBond :753-769 */
                    if (s->ncol == s->capcol) { s->capcol = s->capcol ? 2 * s->capcol : 64; s->col = (cbond*)realloc(s->col, sizeof(cbond) * s->capcol); }
                    cbond* c = &s->col[s->ncol];
                    c->v1 = i1; c->v2 = i2;
                    double E = bond_E(p1->E, p2->E);                                /* LinkVoxels + UpdateConstants */
                    double L = (p1->nom_size + p2->nom_size) * 0.5;
                    c->a1 = E * (L * L) / L;
                    c->f1 = c->f2 = V(0, 0, 0);
                    vox_link_col(p1, s->ncol); vox_link_col(p2, s->ncol);
                    s->ncol++;
                }
            }
        }
    }
}

/* CVX_Sim::UpdateCollisions VX/VX_Sim.cpp:1729-1755 */
static void update_collisions(vxo_sim* s)
{
    s->max_disp_since_update += fabs(s->max_vox_vel * s->dt / s->lat);
    if (s->m.col_system == 2 || s->m.col_system == 3) {
        if (s->max_disp_since_update > (s->m.collision_horizon - 1.0) / 2 || s->col_enable_changed) {
            s->col_enable_changed = 0;
            calc_l1_bonds(s, s->m.collision_horizon);
            s->max_disp_since_update = 0.0;
        }
    } else calc_l1_bonds(s, s->m.collision_horizon);
}

static double ground_penetration(const vxo_sim* s, const voxel* v)                 /* VX/VXS_Voxel.cpp:677-700 (unlimited floor) */
{
    (void)s;
    double Penetration = 0.5 * v->scale - v->pos.z;
    return Penetration <= 0 ? 0 : Penetration;
}

/* CVXS_Voxel::CalcFloorEffect VX/VXS_Voxel.cpp:708-758 */
static v3 floor_effect(vxo_sim* s, voxel* v, v3 TotalVoxForce)
{
    v3 FloorForce = V(0, 0, 0);
    v->static_fric = 0;
    double CurPenetration = ground_penetration(s, v);
    if (CurPenetration > 0) {
        double LocA1 = 2 * v->E * v->nom_size;                                     /* GetLinearStiffness VX/VX_Voxel.h:64 */
        double NormalForce = LocA1 * CurPenetration;
        FloorForce.z += NormalForce;
        FloorForce.z -= s->m.col_damping_z * v->c_lin * v->vel.z;
        double SurfaceVel = sqrt(v->vel.x * v->vel.x + v->vel.y * v->vel.y);
        double SurfaceVelAngle = atan2(v->vel.y, v->vel.x);
        double SurfaceForce = sqrt(TotalVoxForce.x * TotalVoxForce.x + TotalVoxForce.y * TotalVoxForce.y);
        double dFrictionForce = v->u_dynamic * NormalForce;
        v3 FricForceToAdd = vneg(V(cos(SurfaceVelAngle) * dFrictionForce, sin(SurfaceVelAngle) * dFrictionForce, 0));
        if (s->m.variant == 0 && s->m.sticky_floor) { v->vel.x = 0; v->vel.y = 0; v->lin_mom.x = 0; v->lin_mom.y = 0; v->static_fric = 1; }
        if (v->vel.x == 0 && v->vel.y == 0) {
            if (SurfaceForce < v->u_static * NormalForce) v->static_fric = 1;
        } else {
            if (dFrictionForce * s->dt < v->mass * SurfaceVel) FloorForce = vadd(FloorForce, FricForceToAdd);
            else { v->static_fric = 1; v->lin_mom.x = 0; v->lin_mom.y = 0; }
        }
    }
    return FloorForce;
}

/* CVXS_Voxel::CalcTotalForce VX/VXS_Voxel.cpp:482-651 (LW/VXS_Voxel.cpp:296-484) */
static v3 total_force(vxo_sim* s, int vi)
{
    voxel* v = &s->vox[vi];
    int fluid = (s->m.variant == 1 && s->m.fluid_env);
    v3 F = V(0, 0, 0);
    F = vadd(F, vmul(vmul(v->vel, -s->m.slow_damping_z), v->c_lin));
    for (int d = 0; d < 6; d++) {
        if (v->bond[d] < 0) continue;
        const ibond* b = &s->bond[v->bond[d]];
        F = vadd(F, (d % 2) ? b->f2 : b->f1);
    }
    if (s->m.self_col_enabled)
        for (int k = 0; k < v->ncol; k++) { const cbond* c = &s->col[v->col[k]]; F = vadd(F, (c->v2 == vi) ? c->f2 : c->f1); }
    F = vsub(F, V(0, 0, 0));                                                       /* InputForce */
    if (s->m.grav_enabled && !fluid) F.z += v->mass * s->m.grav_acc;
    F = vadd(F, vmul(V(0, 0, 0), 1.0));                                            /* ExternalInputScale*ExternalForce */
    if (fluid) F = vadd(F, v->drag);
    for (int d = 0; d < 6; d++) if (v->bond[d] >= 0) { ibond* b = &s->bond[v->bond[d]]; b->cs_area1 = b->cs_area2 = v->nom_size * v->nom_size; }
    if (s->m.floor_enabled && !fluid) {
        F = vadd(F, floor_effect(s, v, F));
        if (v->static_fric) { F.x = 0; F.y = 0; }
    }
    /* VXS_Voxel.cpp:641-642 (LW :472-475): consumed by the NEXT step's drag mesh */
    v->corner_pos = vdiv(vmul(vadd(V(1, 1, 1), v->strain_pos), v->nom_size), 2);
    v->corner_neg = vdiv(vmul(vneg(vadd(V(1, 1, 1), v->strain_neg)), v->nom_size), 2);
    return F;
}

static v3 total_moment(const vxo_sim* s, const voxel* v)                           /* VX/VXS_Voxel.cpp:653-675 */
{
    v3 M = V(0, 0, 0);
    for (int d = 0; d < 6; d++) {
        if (v->bond[d] < 0) continue;
        const ibond* b = &s->bond[v->bond[d]];
        M = vsub(M, (d % 2) ? b->m2 : b->m1);
    }
    M = vadd(M, vmul(V(0, 0, 0), 1.0));
    return M;
}

/* CVXS_Voxel::EulerStep VX/VXS_Voxel.cpp:169-427 (LW/VXS_Voxel.cpp:148-247) */
static void euler_step(vxo_sim* s, int vi)
{
    voxel* v = &s->vox[vi];
    double dt = s->dt;
    v3 ForceTot = total_force(s, vi);
    v->lin_mom = vadd(v->lin_mom, vmul(ForceTot, dt));
    v3 Disp = vmul(v->lin_mom, dt * v->mass_inv);
    v->pos = vadd(v->pos, Disp);
    v3 Mom = total_moment(s, v);
    v->ang_mom = vadd(v->ang_mom, vmul(Mom, dt));
    double AngMomFact = (1 - 10 * s->m.slow_damping_z * v->inertia_inv * v->c_ang * dt);
    v->ang_mom = V(v->ang_mom.x * AngMomFact, v->ang_mom.y * AngMomFact, v->ang_mom.z * AngMomFact);
    v3 w = vmul(v->ang_mom, v->inertia_inv);
    qt half = Q(0 * 0.5, w.x * 0.5, w.y * 0.5, w.z * 0.5);                         /* 0.5*CQuat(0,w) :195 */
    qt Spin = qmul(half, v->angle);
    v->angle = Q(v->angle.w + Spin.w * dt, v->angle.x + Spin.x * dt, v->angle.y + Spin.y * dt, v->angle.z + Spin.z * dt);
    {   /* NormalizeFast Vec3D.h:243-246 */
        qt* a = &v->angle;
        double l = sqrt(a->x * a->x + a->y * a->y + a->z * a->z + a->w * a->w);
        if (l != 0) { double li = 1.0 / l; a->w *= li; a->x *= li; a->y *= li; a->z *= li; }
        if (a->w >= 1.0) { a->w = 1.0; a->x = 0; a->y = 0; a->z = 0; }
    }

    if (s->m.variant == 0) {
        /* SCALE, VX/VXS_Voxel.cpp:224-340 (the velocity-adjusted development of :343-381 needs NumTimeStepsInWindow > 0,
           which evosoro never writes and the readers refuse) */
        double maxScale = (1 + s->m.growth_amplitude) * v->nom_size;
        double minScale = s->m.min_temp_fact * v->nom_size;
        double currScale;
        double CtrlTempFact = 0, DevTempFact = 0, DevPhaseAddOn = 0, DevTempAmpDampAddOn = 0, k = 0, FrozenTimeAdj = 0, FreezeInitialized = 1;
        double c = (s->cur_time >= 0.5 * s->m.init_cm_time) ? 1.0 : 2 * s->cur_time / s->m.init_cm_time;
        double PreNatalTempFrac = c * ((v->initial_voxel_size / v->nom_size) - 1);
        if (s->m.midlife_freeze_time > 0) {                                         /* :247-264 */
            double middleTime = 0.5 * (s->m.stop_value - s->m.init_cm_time);
            double FreezeStart = middleTime - 0.5 * s->m.midlife_freeze_time;
            double FreezeEnd = middleTime + 0.5 * s->m.midlife_freeze_time;
            if (s->cur_time > FreezeStart && s->cur_time < FreezeEnd) {
                FrozenTimeAdj = s->cur_time - FreezeStart;
                if (s->cur_time < FreezeStart + s->m.init_cm_time) FreezeInitialized = 0;
            }
            if (s->cur_time > FreezeEnd) FrozenTimeAdj = s->m.midlife_freeze_time;
        }
        if (s->cur_time >= v->start_growth_time && v->growth_time > 0) {            /* postnatal linear development :267-296 */
            /* startGrowthTime + growthTime is a FLOAT sum (both members are float) before the double freeze time is added */
            float sg = v->start_growth_time + v->growth_time;
            double EffectiveCurTime = (s->cur_time <= sg + s->m.midlife_freeze_time) ? s->cur_time : sg + s->m.midlife_freeze_time;
            EffectiveCurTime = EffectiveCurTime - FrozenTimeAdj;
            k = (EffectiveCurTime - v->start_growth_time) / v->growth_time;
            if (s->m.final_voxel_size) { float ratio = v->final_voxel_size / v->initial_voxel_size; DevTempFact = k * (ratio - 1.0); }
            if (s->m.final_phase_offset) { float d = v->final_phase_offset - v->phase_offset; DevPhaseAddOn = k * d; }
            if (s->m.final_temp_amp_damp) { float d = v->final_temp_amp_damp - v->temp_amp_damp; DevTempAmpDampAddOn = k * d; }
        }
        if (s->m.temp_enabled && s->cur_time >= s->m.init_cm_time) {
            double ThisCTE = v->cte;
            double ThisPhase = v->phase_offset + DevPhaseAddOn;
            double ThisTempAmpDamp = v->temp_amp_damp + DevTempAmpDampAddOn;
            CtrlTempFact = ThisTempAmpDamp * (v->temp_amplitude * sin(2 * 3.1415926f * (s->cur_time / v->temp_period + ThisPhase))) * ThisCTE * FreezeInitialized;
        }
        if (s->m.initial_voxel_size || s->m.final_voxel_size) {                     /* actuation limited by the current size :316-328 */
            double currSize = (1 + PreNatalTempFrac) * (1 + DevTempFact) * v->nom_size;
            double originalSigmoid = (currSize / v->nom_size - 1) / s->m.growth_amplitude;
            double positiveSigmoid = (originalSigmoid + 1) * 0.5;
            double cappedSigmoid = (positiveSigmoid > 0.5) ? 0.5 : positiveSigmoid;
            CtrlTempFact = CtrlTempFact * cappedSigmoid * 2;
        }
        currScale = CtrlTempFact * v->nom_size + (1 + PreNatalTempFrac) * (1 + DevTempFact) * v->nom_size;
        if (currScale < v->last_scale && currScale < minScale) currScale = v->last_scale;
        if (currScale > v->last_scale && currScale > maxScale) currScale = v->last_scale;
        v->scale = currScale; v->last_scale = v->scale;
    } else {
        /* LW/VXS_Voxel.cpp:211-235 */
        double TempFact = 1.0;
        if (s->m.temp_enabled && s->cur_time >= s->m.init_cm_time)
#ifdef VXO_ANGLE_ADD   /* instrument (libvxoracle_aa.so): the engine's angle-addition form of the actuation sine, see VXO_HALF_ANGLE */
        {
            double a = (double)(2 * 3.1415926f) * (s->cur_time / (double)v->temp_period), b = (double)(2 * 3.1415926f) * (double)v->phase_offset;
            TempFact = (1 + (v->temp_amplitude * (sin(a) * cos(b) + cos(a) * sin(b))) * v->cte);
        }
#else
            TempFact = (1 + (v->temp_amplitude * sin(2 * 3.1415926f * (s->cur_time / v->temp_period + v->phase_offset))) * v->cte);
#endif
        if (TempFact < 0.1) TempFact = 0.1;                                         /* MIN_TEMP_FACTOR LW/VX_Sim.h:31 */
        v->scale = TempFact * v->nom_size;
    }
    v->ang_vel = vmul(v->ang_mom, v->inertia_inv);
    v->vel = vmul(v->lin_mom, v->mass_inv);
}

static v3 get_cm(const vxo_sim* s)                                                 /* VX/VX_Sim.cpp:2415-2430 */
{
    double TotalMass = 0; v3 Sum = V(0, 0, 0);
    for (int i = 0; i < s->nvox; i++) { double m = s->vox[i].mass; Sum = vadd(Sum, vmul(s->vox[i].pos, m)); TotalMass += m; }
    return vdiv(Sum, TotalMass);
}
static double posterior_y(const vxo_sim* s)                                        /* VX/VX_Sim.cpp:2640-2656 */
{
    double r = 100000.0;
    for (int i = 0; i < s->nvox; i++) if (s->vox[i].mat != 5) { double y = s->vox[i].pos.y / s->lat; if (y < r) r = y; }
    return r;
}

/* CVX_Sim::StopConditionMet VX/VX_Sim.cpp:1398-1509 (LW/VX_Sim.cpp:1160-1200: no InitCmTime gate, no afterlife) */
static int stop_condition_met(const vxo_sim* s)
{
    if (s->m.variant == 0 && s->cur_time <= s->m.init_cm_time) return 0;
    switch (s->m.stop_type) {
    case 0: return 0;
    case 1: return s->steps > (int)(s->m.stop_value + 0.5);
    case 2: return s->m.variant == 0 ? s->cur_time > (s->m.stop_value + s->m.afterlife_time) : s->cur_time > s->m.stop_value;
    case 3: return s->cur_time > s->m.temp_period * s->m.stop_value;
    default: return 0;
    }
}

static v3 qrot(qt q, v3 f)                                                         /* RotateVec3D Vec3D.h:293-299 */
{
    double tw = f.x * q.x + f.y * q.y + f.z * q.z;
    double tx = f.x * q.w - f.y * q.z + f.z * q.y;
    double ty = f.x * q.z + f.y * q.w - f.z * q.x;
    double tz = -f.x * q.y + f.y * q.x + f.z * q.w;
    return V(q.w * tx + q.x * tw + q.y * tz - q.z * ty, q.w * ty - q.x * tz + q.y * tw + q.z * tx, q.w * tz + q.x * ty - q.y * tx + q.z * tw);
}
static v3 vcross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static double vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 vnormalized(v3 a) { double l = sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return l > 0 ? vdiv(a, l) : a; }   /* Vec3D.h:127 */

/* fluid drag, LW/VX_Sim.cpp:1516-1597 with UpdateMeshPhysicsOnlyNoColors / GetCurVLoc (LW/VX_MeshUtil.cpp:368-428) and
 * CalcFaceNormals (LW/Utils/Mesh.cpp:659-665) */
static void fluid_drag(vxo_sim* s)
{
    for (int i = 0; i < s->nvox; i++) s->vox[i].drag = V(0, 0, 0);
    for (int i = 0; i < s->nmv; i++) {
        struct mvert* m = &s->mv[i];
        v3 avg = V(0, 0, 0); double tw = 0;
        for (int j = 0; j < m->nc; j++) {
            const voxel* v = &s->vox[m->vox[j]];
            v3 cn = v->corner_neg, cp = v->corner_pos, off;
            switch (m->corner[j]) {                                                /* NNN..PPP, LW/VX_MeshUtil.h:23-30 */
            case 0: off = cn; break;
            case 1: off = V(cn.x, cn.y, cp.z); break;
            case 2: off = V(cn.x, cp.y, cn.z); break;
            case 3: off = V(cn.x, cp.y, cp.z); break;
            case 4: off = V(cp.x, cn.y, cn.z); break;
            case 5: off = V(cp.x, cn.y, cp.z); break;
            case 6: off = V(cp.x, cp.y, cn.z); break;
            default: off = cp; break;
            }
            v3 p = vadd(v->pos, qrot(v->angle, off));
            avg = vadd(avg, vmul(p, 1.0)); tw += 1.0;
        }
        v3 np = vdiv(avg, tw);
        v3 draw = vsub(np, m->v0);                                                 /* DrawOffset */
        m->cur = vadd(m->v0, draw);                                                /* OffPos() */
    }
    for (int i = 0; i < s->nmf; i++) {
        struct mfacet* f = &s->mf[i];
        f->n = vnormalized(vcross(vsub(s->mv[f->vi[1]].cur, s->mv[f->vi[0]].cur), vsub(s->mv[f->vi[2]].cur, s->mv[f->vi[0]].cur)));
    }
    for (int i = 0; i < s->nmf; i++) {
        struct mfacet* f = &s->mf[i];
        v3 A = s->mv[f->vi[0]].cur, B = s->mv[f->vi[1]].cur, C = s->mv[f->vi[2]].cur;
        v3 AB = vsub(B, A), AC = vsub(C, A);
        double area = fabs(vlen(vcross(AB, AC)) / 2.0);
        v3 speed = s->vox[f->owner].vel, n = f->n, proj = V(0, 0, 0);
        float ang = (float)acos(vdot(vnormalized(speed), vnormalized(n)));
        if (fabsf(ang) < VX_PI / 2) proj = vmul(vnormalized(n), vdot(speed, n));       /* ProjectOnTo, LW/Utils/Vec3D.h:142 */
        v3 drag = vmul(vnormalized(proj), -s->m.aggregate_drag_coef * area * vlen2(proj));
        s->vox[f->owner].drag = vadd(s->vox[f->owner].drag, drag);
    }
}

/* LinkSimVoxels, LW/VX_MeshUtil.cpp:110-276: vertices = lattice corners touched by 1..7 voxels, two triangles per
 * exposed voxel face (owner = that voxel), vertex rest positions with the 1e-6 offset hack of GetXYZ (LW/VX_Object.cpp:521-540) */
static void build_drag_mesh(vxo_sim* s, const int* x2s)
{
    const vxo_model* m = &s->m;
    int tx = m->nx + 1, ty = m->ny + 1, tz = m->nz + 1, nall = tx * ty * tz;
    struct mvert* all = (struct mvert*)calloc(nall, sizeof(struct mvert));
    int* map = (int*)malloc(sizeof(int) * nall);
    s->mf = (struct mfacet*)calloc((size_t)12 * s->nvox + 1, sizeof(struct mfacet));
    static const int cdx[8] = {0, 0, 0, 0, 1, 1, 1, 1}, cdy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, cdz[8] = {0, 1, 0, 1, 0, 1, 0, 1};
#define D3(X, Y, Z) ((Z) * tx * ty + (Y) * tx + (X))
#define OCC(X, Y, Z) ((X) >= 0 && (Y) >= 0 && (Z) >= 0 && (X) < m->nx && (Y) < m->ny && (Z) < m->nz && m->structure[(X) + m->nx * (Y) + m->nx * m->ny * (Z)] != 0)
#define FACET(a, b, c) do { struct mfacet* f = &s->mf[s->nmf++]; f->vi[0] = (a); f->vi[1] = (b); f->vi[2] = (c); f->owner = vi; } while (0)
    int ncell = m->nx * m->ny * m->nz;
    for (int i = 0, vi = 0; i < ncell; i++) {
        if (m->structure[i] == 0) continue;
        int cz = i / (m->nx * m->ny), cy = (i - cz * m->nx * m->ny) / m->nx, cx = i - cz * m->nx * m->ny - cy * m->nx;
        for (int c = 0; c < 8; c++) { struct mvert* v = &all[D3(cx + cdx[c], cy + cdy[c], cz + cdz[c])]; v->vox[v->nc] = vi; v->corner[v->nc] = c; v->nc++; }
        if (!OCC(cx + 1, cy, cz)) { FACET(D3(cx + 1, cy, cz), D3(cx + 1, cy + 1, cz), D3(cx + 1, cy + 1, cz + 1)); FACET(D3(cx + 1, cy, cz), D3(cx + 1, cy + 1, cz + 1), D3(cx + 1, cy, cz + 1)); }
        if (!OCC(cx - 1, cy, cz)) { FACET(D3(cx, cy, cz), D3(cx, cy + 1, cz + 1), D3(cx, cy + 1, cz)); FACET(D3(cx, cy, cz), D3(cx, cy, cz + 1), D3(cx, cy + 1, cz + 1)); }
        if (!OCC(cx, cy + 1, cz)) { FACET(D3(cx, cy + 1, cz), D3(cx, cy + 1, cz + 1), D3(cx + 1, cy + 1, cz + 1)); FACET(D3(cx, cy + 1, cz), D3(cx + 1, cy + 1, cz + 1), D3(cx + 1, cy + 1, cz)); }
        if (!OCC(cx, cy - 1, cz)) { FACET(D3(cx, cy, cz), D3(cx + 1, cy, cz + 1), D3(cx, cy, cz + 1)); FACET(D3(cx, cy, cz), D3(cx + 1, cy, cz), D3(cx + 1, cy, cz + 1)); }
        if (!OCC(cx, cy, cz + 1)) { FACET(D3(cx, cy, cz + 1), D3(cx + 1, cy, cz + 1), D3(cx + 1, cy + 1, cz + 1)); FACET(D3(cx, cy, cz + 1), D3(cx + 1, cy + 1, cz + 1), D3(cx, cy + 1, cz + 1)); }
        if (!OCC(cx, cy, cz - 1)) { FACET(D3(cx, cy, cz), D3(cx + 1, cy + 1, cz), D3(cx + 1, cy, cz)); FACET(D3(cx, cy, cz), D3(cx, cy + 1, cz), D3(cx + 1, cy + 1, cz)); }
        vi++;
    }
    (void)x2s;
    s->mv = (struct mvert*)calloc(nall, sizeof(struct mvert));
    double lat = m->lattice_dim, half = (1.0 / 2) * lat;                            /* GetLatDimEnv()/2 */
    for (int k = 0, idx = 0; k < tz; k++) for (int j = 0; j < ty; j++) for (int i = 0; i < tx; i++, idx++) {
        map[idx] = -1;
        if (all[idx].nc != 0 && all[idx].nc != 8) {
            double Eps = 0.000001;
            v3 p = V(lat * (1.0 * (0.5 + i) + Eps), lat * (1.0 * (0.5 + j) + Eps), lat * 1.0 * (0.5 + k));
            all[idx].v0 = vsub(p, V(half, half, half));
            s->mv[s->nmv] = all[idx];
            map[idx] = s->nmv++;
        }
    }
    for (int f = 0; f < s->nmf; f++) for (int j = 0; j < 3; j++) s->mf[f].vi[j] = map[s->mf[f].vi[j]];
    free(all); free(map);
#undef D3
#undef OCC
#undef FACET
}

/* CVX_Sim::TimeStep + Integrate + UpdateStats, VX/VX_Sim.cpp:1054-1156,1763-1933,1511-1690 */
static int time_step(vxo_sim* s)
{
    if (!s->cm_init && s->cur_time > s->m.init_cm_time) { s->ini_cm = s->cur_cm; s->cm_init = 1; }
    if (s->m.variant == 0 && s->cur_time >= s->m.stop_value && s->end_of_life_posterior_y == 0)
        s->end_of_life_posterior_y = posterior_y(s);
    if (s->m.self_col_enabled) update_collisions(s);
    else if (s->col_enable_changed) { s->col_enable_changed = 0; }

    int diverged = 0;
    for (int i = 0; i < s->nbond; i++) { bond_update(s, &s->bond[i]); if (s->bond[i].strain_tot > 100) diverged = 1; }
    if (diverged) return 0;
    for (int i = 0; i < s->ncol; i++) col_update(s, &s->col[i]);
    s->dt = s->m.dt_frac * s->opt_dt;
    if (s->m.variant == 1 && s->m.fluid_env) fluid_drag(s);
    for (int i = 0; i < s->nvox; i++) euler_step(s, i);
    s->cur_time += s->dt;
    s->steps++;
    /* UpdateStats: CoM and (collisions only) max velocity */
    s->cur_cm = get_cm(s);
    if (s->m.variant == 0 && s->m.time_between_traces > 0 && s->cur_time > s->m.init_cm_time) {     /* VX_Sim.cpp:1537-1547 */
        if (s->ntrace == 0 || s->trace[4 * (s->ntrace - 1)] + s->m.time_between_traces <= s->cur_time) {
            if (s->ntrace == s->captrace) { s->captrace = s->captrace ? 2 * s->captrace : 64; s->trace = (double*)realloc(s->trace, sizeof(double) * 4 * s->captrace); }
            double* e = s->trace + 4 * s->ntrace++;
            e[0] = s->cur_time; e[1] = s->cur_cm.x; e[2] = s->cur_cm.y; e[3] = s->cur_cm.z;
        }
    }
    {
        double mv2 = 0;
        for (int i = 0; i < s->nvox; i++) { double t = vlen2(s->vox[i].vel); if (t > mv2) mv2 = t; }
        s->max_vox_vel = sqrt(mv2);
    }
    return 1;
}

/* CVX_Sim::CalcMaxDt VX/VX_Sim.cpp:1693-1727 */
static double calc_max_dt(const vxo_sim* s)
{
    double MaxFreq2 = 0;
    if (s->nbond != 0) {
        for (int i = 0; i < s->nbond; i++) {
            const ibond* b = &s->bond[i];
            if (b->a1 / s->vox[b->v1].mass > MaxFreq2) MaxFreq2 = b->a1 / s->vox[b->v1].mass;
            if (b->a1 / s->vox[b->v2].mass > MaxFreq2) MaxFreq2 = b->a1 / s->vox[b->v2].mass;
        }
    } else {
        if (s->nvox == 0) return 0;
        for (int i = 0; i < s->nvox; i++) if (s->vox[i].E / s->vox[i].mass > MaxFreq2) MaxFreq2 = s->vox[i].E / s->vox[i].mass;
    }
    double MaxFreq = sqrt(MaxFreq2);
    return 1.0 / (MaxFreq * 2 * (double)3.1415926);
}

static int cmp_int(const void* a, const void* b) { int x = *(const int*)a, y = *(const int*)b; return (x > y) - (x < y); }

/* CVX_Voxel::CalcNearby VX/VX_Voxel.cpp:171-209: every voxel reachable in <= hops bond hops (self included) */
static void calc_nearby(vxo_sim* s, int vi, int hops, int* scratch, unsigned char* mark)
{
    int n = 0, start = 0, stop = 1;
    scratch[n++] = vi; mark[vi] = 1;
    for (int h = 0; h < hops; h++) {
        for (int j = start; j < stop; j++) {
            const voxel* t = &s->vox[scratch[j]];
            for (int k = 0; k < 6; k++) if (t->bond[k] >= 0) {
                const ibond* b = &s->bond[t->bond[k]];
                int other = (k % 2) ? b->v1 : b->v2;
                if (!mark[other]) { mark[other] = 1; scratch[n++] = other; }
            }
        }
        start = stop; stop = n;
    }
    voxel* v = &s->vox[vi];
    v->nnear = n; v->near = (int*)malloc(sizeof(int) * n);
    memcpy(v->near, scratch, sizeof(int) * n);
    for (int i = 0; i < n; i++) mark[scratch[i]] = 0;
    qsort(v->near, n, sizeof(int), cmp_int);
}

/* CVX_Sim::Import + ResetSimulation, VX/VX_Sim.cpp:488-717,839-1007 (LW/VX_Sim.cpp: SetVoxData before bonds) */
vxo_sim* vxo_create(const vxo_model* m)
{
    init_consts();
    vxo_sim* s = (vxo_sim*)calloc(1, sizeof(vxo_sim));
    s->m = *m;
    s->lat = m->lattice_dim;
    int ncell = m->nx * m->ny * m->nz;
    int* x2s = (int*)malloc(sizeof(int) * (ncell > 0 ? ncell : 1));
    for (int i = 0; i < ncell; i++) { x2s[i] = -1; if (m->structure[i] != 0) s->nvox++; }
    s->max_disp_since_update = (double)FLT_MAX;                                    /* ClearAll :369 */
    s->col_enable_changed = 1;                                                     /* ctor :50 / EnableFeature :412 */
    if (s->nvox == 0) { s->status = 3; free(x2s); return s; }
    s->vox = (voxel*)calloc(s->nvox, sizeof(voxel));
    double scale = m->lattice_dim * 1.0;                                           /* GetLatDimEnv().x, VX/VX_Object.h:377 */
    int si = 0;
    for (int i = 0; i < ncell; i++) {
        if (m->structure[i] == 0) continue;
        int iz = i / (m->nx * m->ny), iy = (i - iz * m->nx * m->ny) / m->nx, ix = i - iz * m->nx * m->ny - iy * m->nx; /* GetXYZNom */
        voxel* v = &s->vox[si];
        voxel_set_material(m, v, m->structure[i], scale);
        v->nom_pos = V(scale * (ix + 0.5), scale * (iy + 0.5), scale * (iz + 0.5));  /* GetXYZ :543 */
        for (int d = 0; d < 6; d++) v->bond[d] = -1;
        x2s[i] = si++;
    }
    /* LW applies the evolved stiffness before bonds exist (LW/VX_Sim.cpp SetVoxData) and refreshes damping terms */
    if (m->variant == 1 && m->stiffness)
        for (int i = 0; i < s->nvox; i++) {
            voxel* v = &s->vox[i];
            v->E = m->stiffness[i];
            v->c_lin = 2 * sqrt(v->mass * v->E * v->nom_size);
            v->c_ang = 2 * sqrt(v->inertia * v->E * v->nom_size * v->nom_size * v->nom_size);
        }
    /* permanent bonds: for each voxel, +X then +Y then +Z neighbour (:623-641) */
    s->bond = (ibond*)calloc((size_t)3 * s->nvox + 1, sizeof(ibond));
    for (int i = 0, vi = 0; i < ncell; i++) {
        if (m->structure[i] == 0) continue;
        int iz = i / (m->nx * m->ny), iy = (i - iz * m->nx * m->ny) / m->nx, ix = i - iz * m->nx * m->ny - iy * m->nx;
        for (int j = 0; j < 3; j++) {
            int px = ix + (j == 0), py = iy + (j == 1), pz = iz + (j == 2);
            if (px >= m->nx || py >= m->ny || pz >= m->nz) continue;
            int pi = px + m->nx * py + m->nx * m->ny * pz;
            if (!m->structure[pi]) continue;
            int other = x2s[pi];
            if (s->vox[vi].E == 0 || s->vox[other].E == 0) continue;                /* LinkVoxels fails, VX_Bond.cpp:84 */
            bond_link(s, &s->bond[s->nbond], vi, other, j + 1);
            s->vox[vi].bond[2 * j] = s->nbond; s->vox[other].bond[2 * j + 1] = s->nbond;
            s->nbond++;
        }
        vi++;
    }
    if (m->variant == 1 && m->fluid_env) build_drag_mesh(s, x2s);
    free(x2s);
    /* surface list + nearby lists (:649-659) */
    s->surf = (int*)malloc(sizeof(int) * s->nvox);
    int* scratch = (int*)malloc(sizeof(int) * s->nvox);
    unsigned char* mark = (unsigned char*)calloc(s->nvox, 1);
    for (int i = 0; i < s->nvox; i++) {
        int surface = 0;
        for (int d = 0; d < 6; d++) if (s->vox[i].bond[d] < 0) surface = 1;
        if (surface) s->surf[s->nsurf++] = i;
        calc_nearby(s, i, (int)(m->collision_horizon * 1.5), scratch, mark);
    }
    free(scratch); free(mark);
    /* ResetSimulation: ResetVoxel (VX/VXS_Voxel.cpp:98-137) then per-voxel float parameters (:878-991) */
    for (int i = 0; i < s->nvox; i++) {
        voxel* v = &s->vox[i];
        v->pos = v->nom_pos; v->angle = Q(1.0, 0, 0, 0);
        v->scale = v->nom_size; v->last_scale = v->scale;
        v->corner_pos = V(v->scale / 2, v->scale / 2, v->scale / 2);               /* ResetVoxel VXS_Voxel.cpp:131-132 */
        v->corner_neg = V(-v->scale / 2, -v->scale / 2, -v->scale / 2);
        v->temp_amplitude = (float)m->temp_amplitude;
        v->temp_period = (float)m->temp_period;
        v->phase_offset = m->phase_offset ? (float)m->phase_offset[i] : (float)0.0;
        v->temp_amp_damp = m->temp_amp_damp ? (float)m->temp_amp_damp[i] : (float)1.0;
        /* development parameters, VX/VX_Sim.cpp:885-975; every member is a float, the right-hand sides are doubles */
        v->final_phase_offset = m->final_phase_offset ? (float)m->final_phase_offset[i] : (float)0.0;
        v->final_temp_amp_damp = m->final_temp_amp_damp ? (float)m->final_temp_amp_damp[i] : (float)1.0;
        v->onset_bound = (float)m->stop_value;            /* OnsetRelative / TerminationRelative (ParentLifetime) are refused by the readers */
        v->termination_bound = (float)m->stop_value;
        if (m->initial_voxel_size) {
            double tf = 1 + (m->growth_amplitude * m->initial_voxel_size[i]);
            double eff = (tf < m->min_temp_fact) ? m->min_temp_fact : tf;
            v->initial_voxel_size = (float)(eff * v->nom_size);
        } else v->initial_voxel_size = (float)v->nom_size;
        if (m->final_voxel_size) {
            double tf = 1 + (m->growth_amplitude * m->final_voxel_size[i]);
            double eff = (tf < m->min_temp_fact) ? m->min_temp_fact : tf;
            v->final_voxel_size = (float)(eff * v->nom_size);
        } else v->final_voxel_size = v->initial_voxel_size;
        if (m->start_growth_time) {
            double t0 = m->start_growth_time[i] * (v->onset_bound - m->init_cm_time) + m->init_cm_time;
            if (t0 >= v->onset_bound - m->min_growth_time) v->start_growth_time = (float)(v->onset_bound - m->min_growth_time);
            else v->start_growth_time = (float)t0;
        } else if (m->final_voxel_size || m->growth_time) v->start_growth_time = (float)m->init_cm_time;
        else v->start_growth_time = (float)(m->stop_value - m->midlife_freeze_time);
        if (m->growth_time) {
            float span = v->termination_bound - v->start_growth_time;                  /* float - float */
            double g = m->growth_time[i] * (span - m->midlife_freeze_time);
            v->growth_time = (float)((g <= m->min_growth_time) ? m->min_growth_time : g);
        } else if (m->final_voxel_size) {
            float span = v->termination_bound - v->start_growth_time;
            v->growth_time = (float)(span - m->midlife_freeze_time);
        } else v->growth_time = (float)m->min_growth_time;
        if (m->variant == 0 && m->stiffness) v->E = m->stiffness[i];               /* SetEMod without refresh, :983-988 */
    }
    s->opt_dt = calc_max_dt(s);
    return s;
}

void vxo_destroy(vxo_sim* s)
{
    if (!s) return;
    for (int i = 0; i < s->nvox; i++) { free(s->vox[i].col); free(s->vox[i].near); }
    free(s->vox); free(s->bond); free(s->surf); free(s->col); free(s->mv); free(s->mf); free(s->trace); free(s);
}

/* the loop of voxelyzeMain/main.cpp:89-111; a diverged robot would spin forever there, we stop and flag it */
long vxo_step(vxo_sim* s, long max_steps)
{
    long n = 0;
    if (s->status == 3 || s->status == 2) return 0;
    while ((max_steps < 0 || n < max_steps)) {
        if (stop_condition_met(s)) { s->status = 1; break; }
        if (!time_step(s)) { s->status = 2; break; }
        n++;
    }
    return n;
}

void vxo_get_info(const vxo_sim* s, vxo_info* o)
{
    memset(o, 0, sizeof(*o));
    o->nvox = s->nvox; o->nbond = s->nbond; o->nsurf = s->nsurf; o->ncol = s->ncol; o->steps = s->steps; o->status = s->status;
    o->cm_initialized = s->cm_init; o->col_rebuilds = s->rebuilds;
    for (int i = 0; i < s->nbond; i++) o->n_small_angle += s->bond[i].small_angle;
    o->opt_dt = s->opt_dt; o->dt = s->dt; o->cur_time = s->cur_time; o->max_vox_vel = s->max_vox_vel;
    o->cur_cm[0] = s->cur_cm.x; o->cur_cm[1] = s->cur_cm.y; o->cur_cm[2] = s->cur_cm.z;
    o->ini_cm[0] = s->ini_cm.x; o->ini_cm[1] = s->ini_cm.y; o->ini_cm[2] = s->ini_cm.z;
}

void vxo_get_state(const vxo_sim* s, double* o)
{
    for (int i = 0; i < s->nvox; i++) {
        const voxel* v = &s->vox[i]; double* r = o + 14 * (size_t)i;
        r[0] = v->pos.x; r[1] = v->pos.y; r[2] = v->pos.z;
        r[3] = v->angle.w; r[4] = v->angle.x; r[5] = v->angle.y; r[6] = v->angle.z;
        r[7] = v->scale; r[8] = v->vel.x; r[9] = v->vel.y; r[10] = v->vel.z;
        r[11] = v->ang_vel.x; r[12] = v->ang_vel.y; r[13] = v->ang_vel.z;
    }
}

void vxo_get_bond_table(const vxo_sim* s, int* v1, int* v2, int* axis)
{
    for (int i = 0; i < s->nbond; i++) { v1[i] = s->bond[i].v1; v2[i] = s->bond[i].v2; axis[i] = s->bond[i].axis; }
}

/* instrument: the SmallAngle flag of every bond (1 = small-angle branch of CalcLinForce), in bond-table order */
void vxo_get_bond_modes(const vxo_sim* s, int* small)
{
    for (int i = 0; i < s->nbond; i++) small[i] = s->bond[i].small_angle;
}

/* the constants Import leaves on every voxel (SetMaterial, VX_Voxel.cpp:94-128) and bond (UpdateConstants, VX_Bond.cpp:95-173), for the
 * bit-for-bit comparison with the product's host model (tests/test_capi.py): 12 doubles per voxel, 23 per bond */
void vxo_get_constants(const vxo_sim* s, double* vox12n, double* bond23n)
{
    for (int i = 0; i < s->nvox; i++) {
        const voxel* v = &s->vox[i]; double* o = vox12n + 12 * i;
        o[0] = v->mass; o[1] = v->mass_inv; o[2] = v->inertia; o[3] = v->inertia_inv; o[4] = v->first_moment; o[5] = v->c_lin; o[6] = v->c_ang;
        o[7] = v->E; o[8] = v->nom_size; o[9] = v->u_static; o[10] = v->u_dynamic; o[11] = v->cte;
    }
    for (int i = 0; i < s->nbond; i++) {
        const ibond* b = &s->bond[i]; double* o = bond23n + 23 * i;
        o[0] = b->v1; o[1] = b->v2; o[2] = b->axis - 1; o[3] = b->homogeneous; o[4] = b->L; o[5] = b->a1; o[6] = b->a2;
        o[7] = b->b1y; o[8] = b->b2y; o[9] = b->b3y; o[10] = b->b1z; o[11] = b->b2z; o[12] = b->b3z;
        o[13] = b->sq_a1m1; o[14] = b->sq_a1m2; o[15] = b->sq_a2i1; o[16] = b->sq_a2i2;
        o[17] = b->sq_b1ym1; o[18] = b->sq_b1ym2; o[19] = b->sq_b2yfm1; o[20] = b->sq_b2yfm2; o[21] = b->sq_b3yi1; o[22] = b->sq_b3yi2;
    }
}

/* Test instrument, not part of the restatement: moves every position and quaternion component of every voxel to the next representable double
 * above or below (pseudo-random choice from `seed`).  tests/test_gpu_ledger.py applies it before every step of a twin run to
 * measure how far the reference algorithm itself drifts under rounding-size noise in its state -- the size of noise any
 * re-association or fused multiply-add of its arithmetic introduces at every step. */
void vxo_jitter(vxo_sim* s, unsigned seed)
{
    unsigned long long x = 0x9E3779B97F4A7C15ull ^ ((unsigned long long)seed * 0xD1B54A32D192ED03ull);
    for (int i = 0; i < s->nvox; i++) {
        double* c[7] = {&s->vox[i].pos.x, &s->vox[i].pos.y, &s->vox[i].pos.z, &s->vox[i].angle.w, &s->vox[i].angle.x, &s->vox[i].angle.y, &s->vox[i].angle.z};
        for (int k = 0; k < 7; k++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            *c[k] = nextafter(*c[k], (x & 1) ? INFINITY : -INFINITY);
        }
    }
}

/* test instrument (scripts/dev_gpu_diag.py drift7): the voxels' state replaced by the one given, in the layout of vxo_get_state (momenta
 * from the velocities); bond histories and modes stay.  One step from the ENGINE's state shows what a single step of the two
 * implementations differs by, without the history of earlier differences. */
void vxo_set_state(vxo_sim* s, const double* in14n)
{
    for (int i = 0; i < s->nvox; i++) {
        voxel* v = &s->vox[i]; const double* r = in14n + 14 * (size_t)i;
        v->pos = V(r[0], r[1], r[2]);
        v->angle.w = r[3]; v->angle.x = r[4]; v->angle.y = r[5]; v->angle.z = r[6];
        v->scale = r[7];
        v->vel = V(r[8], r[9], r[10]); v->ang_vel = V(r[11], r[12], r[13]);
        v->lin_mom = vmul(v->vel, v->mass); v->ang_mom = vmul(v->ang_vel, v->inertia);
    }
}

int vxo_get_cm_trace(const vxo_sim* s, double* out4n, int capacity)
{
    for (int i = 0; i < s->ntrace && i < capacity; i++) memcpy(out4n + 4 * i, s->trace + 4 * i, 4 * sizeof(double));
    return s->ntrace;
}

double vxo_alg_bytes_per_step(const vxo_sim* s) { return 224.0 * s->nvox + 144.0 * s->nbond; }

/* CVX_SimGA::WriteResultFile VX/VX_SimGA.cpp:33-203 (LW/VX_SimGA.cpp:33-77) */
void vxo_get_result(const vxo_sim* s, vxo_result* r)
{
    memset(r, 0, sizeof(*r));
    r->status = s->status; r->steps = s->steps; r->nvox = s->nvox; r->nbond = s->nbond;
    r->dt = s->dt; r->cur_time = s->cur_time;
    r->ini_cm[0] = s->ini_cm.x; r->ini_cm[1] = s->ini_cm.y; r->ini_cm[2] = s->ini_cm.z;
    r->cur_cm[0] = s->cur_cm.x; r->cur_cm[1] = s->cur_cm.y; r->cur_cm[2] = s->cur_cm.z;
    double lat = s->lat;
    if (s->m.variant == 0) {
        r->lifetime = s->cur_time - s->m.afterlife_time;
        double finalDist = pow(pow(s->cur_cm.x - s->ini_cm.x, 2) + pow(s->cur_cm.y - s->ini_cm.y, 2), 0.5) / lat;
        /* SS.* distances as left by the last UpdateStats (VX/VX_Sim.cpp:1520-1534, 2584-2712) */
        double ant = 0.0, post = 100000.0, anty = 0.0, posty = 100000.0; int touching = 0, feet = 0;
        for (int i = 0; i < s->nvox; i++) {
            const voxel* v = &s->vox[i];
            double d = pow(pow(v->pos.x - s->ini_cm.x, 2) + pow(v->pos.y - s->ini_cm.y, 2), 0.5) / lat;
            if (d > ant) ant = d;
            if (d < post) post = d;
            if (v->mat != 5) { double y = v->pos.y / lat; if (y > anty) anty = y; if (y < posty) posty = y; }
            if (ground_penetration(s, v) > 0) { touching++; if (v->mat == 6) feet++; }
        }
        if (s->steps == 0) { ant = post = anty = posty = 0; touching = feet = 0; }   /* SS.Clear() values */
        r->final_dist = finalDist; r->norm_final_dist = finalDist - 0; r->norm_frozen_dist = 0;
        r->norm_regime_dist = post - s->end_of_life_posterior_y;
        r->final_dist_y = (s->cur_cm.y - s->ini_cm.y) / lat;
        r->anterior_dist = ant; r->posterior_dist = post; r->anterior_y = anty; r->posterior_y = posty;
        r->end_of_life_posterior_y = s->end_of_life_posterior_y; r->fall_adj_post_y = s->end_of_life_posterior_y;
        r->num_non_feet_touching_floor = feet; r->num_touching_floor = touching;
    } else {
        r->lifetime = s->cur_time;
        v3 d = vdiv(vsub(s->cur_cm, s->ini_cm), lat);                               /* float-typed in the reference */
        r->norm_dist_x = (float)d.x; r->norm_dist_y = (float)d.y; r->norm_dist_z = (float)d.z;
        r->norm_abs_disp = (float)vlen(d);
    }
}
