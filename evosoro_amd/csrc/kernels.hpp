// HIP kernels of the batched Voxelyze time-stepper (gfx950).  FP64 throughout, SoA state in HBM
// (device_types.hpp).  Reference loops being replaced (paths under evosoro/_voxcad/Voxelyze/):
//   step_control    CVX_Sim::TimeStep prologue + StopConditionMet + UpdateCollisions and the `CurTime += dt`
//                   epilogue (VX_Sim.cpp:1054-1135,1398-1423,1729-1755,1929)
//   rebuild_rows    CVX_Sim::CalcL1Bonds (VX_Sim.cpp:2357-2413)
//   bond_compute    CVXS_BondInternal::CalcLinForce / UpdateBondStrain / AddDampForces (VXS_BondInternal.cpp:56-346)
//   voxel_update    CVXS_Voxel::EulerStep / CalcTotalForce / CalcTotalMoment / CalcFloorEffect (VXS_Voxel.cpp:169-758),
//                   CVXS_BondCollision::CalcContactForce (VXS_BondCollision.cpp:41-59), MaxVoxVel of UpdateStats
// Two launch shapes share those device functions:
//   k_robot_steps<BLOCK>  fused path: ONE workgroup per robot (robots up to BLOCK voxels), thread = voxel + its
//                         three positive bonds; Force2/Moment2 of every bond travel to the neighbour voxel through
//                         LDS, so per step only voxel state and bond history cross HBM; several steps per launch.
//   k_step_begin / k_bonds / k_voxels   streaming path for lattices of any size (one thread per bond slot / voxel,
//                         bond outputs through HBM).
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.hpp"

namespace vxh {

// SoA component planes behind three base pointers (few SGPRs): see DBatch in device_types.hpp
#define VS(comp, v) B.vs[(unsigned)(comp) * (unsigned)B.nv + (unsigned)(v)]
#define POS(cur, k, v) VS((cur) * 4 + (k), v)
#define SCALE(cur, v) VS((cur) * 4 + 3, v)
#define QUAT(k, v) VS(8 + (k), v)
#define LINMOM(k, v) VS(12 + (k), v)
#define ANGMOM(k, v) VS(15 + (k), v)
#define HIST(k, slot) B.hist[(unsigned)(k) * 3u * (unsigned)B.nv + (unsigned)(slot)]
#define BOUT(k, slot) B.bout[(unsigned)(k) * 3u * (unsigned)B.nv + (unsigned)(slot)]

struct d3 { double x, y, z; };
struct dq { double w, x, y, z; };

__device__ __forceinline__ d3 mk3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ dq mkq(double w, double x, double y, double z) { dq r; r.w = w; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3 operator-(d3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ d3 operator*(d3 a, double f) { return mk3(f * a.x, f * a.y, f * a.z); }
__device__ __forceinline__ double len2(d3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ dq conj(dq a) { return mkq(a.w, -a.x, -a.y, -a.z); }
__device__ __forceinline__ dq qmul(dq a, dq f)      // Vec3D.h:193
{
    return mkq(a.w * f.w - a.x * f.x - a.y * f.y - a.z * f.z,
               a.w * f.x + a.x * f.w + a.y * f.z - a.z * f.y,
               a.w * f.y - a.x * f.z + a.y * f.w + a.z * f.x,
               a.w * f.z + a.x * f.y - a.y * f.x + a.z * f.w);
}
__device__ __forceinline__ d3 rotinv(dq q, d3 f)    // CQuat::RotateVec3DInv, Vec3D.h:300-314
{
    double tw = q.x * f.x + q.y * f.y + q.z * f.z;
    double tx = q.w * f.x - q.y * f.z + q.z * f.y;
    double ty = q.w * f.y + q.x * f.z - q.z * f.x;
    double tz = q.w * f.z - q.x * f.y + q.y * f.x;
    return mk3(tw * q.x + tx * q.w + ty * q.z - tz * q.y,
               tw * q.y - tx * q.z + ty * q.w + tz * q.x,
               tw * q.z + tx * q.y - ty * q.x + tz * q.w);
}
// ToXDirBond / ToOrigDirBond, VX_Bond.h:45-48 ; axis 0 = X, 1 = Y, 2 = Z
__device__ __forceinline__ d3 to_xdir(int axis, d3 p) { return axis == 1 ? mk3(p.y, -p.x, p.z) : (axis == 2 ? mk3(p.z, p.y, -p.x) : p); }
__device__ __forceinline__ dq to_xdir(int axis, dq q) { return axis == 1 ? mkq(q.w, q.y, -q.x, q.z) : (axis == 2 ? mkq(q.w, q.z, q.y, -q.x) : q); }
__device__ __forceinline__ d3 to_orig(int axis, d3 p) { return axis == 1 ? mk3(-p.y, p.x, p.z) : (axis == 2 ? mk3(-p.z, p.y, p.x) : p); }

#define VXH_PI 3.14159265358979
#define VXH_DISCARD_ANGLE_RAD 1e-7
#define VXH_SMALL_ANGLE_RAD 1.732e-2
#define VXH_SA_BOND_BEND_RAD 0.05
#define VXH_SA_BOND_EXT_PERC 1.30
#define VXH_HYST 1.1

// CQuat::FromAngleToPosX, Vec3D.h:208-237
__device__ __forceinline__ dq from_angle_to_pos_x(d3 from)
{
    if (from.x == 0 && from.y == 0 && from.z == 0) return mkq(1, 0, 0, 0);
    double yox = from.y / from.x, zox = from.z / from.x;
    if (yox < VXH_SMALL_ANGLE_RAD && yox > -VXH_SMALL_ANGLE_RAD && zox < VXH_SMALL_ANGLE_RAD && zox > -VXH_SMALL_ANGLE_RAD) {
        double y = 0.5 * zox, z = -0.5 * yox;
        return mkq(1 + 0.5 * (-y * y - z * z), 0, y, z);
    }
    double l = sqrt(from.x * from.x + from.y * from.y + from.z * from.z);
    d3 n = from;
    if (l > 0) { double li = 1.0 / l; n.x *= li; n.y *= li; n.z *= li; }
    double theta = acos(n.x);
    if (theta > VXH_PI - VXH_DISCARD_ANGLE_RAD) return mkq(0, 0, 1, 0);
    double axis_inv = 1.0 / sqrt(n.z * n.z + n.y * n.y);
    double a = 0.5 * theta, s, c;
    sincos(a, &s, &c);
    return mkq(c, 0, n.z * axis_inv * s, -n.y * axis_inv * s);
}
// CQuat::ToRotationVector, Vec3D.h:270-285
__device__ __forceinline__ d3 to_rotvec(dq q, double slthresh)
{
    double sl = 1.0 - q.w * q.w;
    if (sl <= 0) return mk3(0, 0, 0);
    double wc = q.w > 1 ? 1 : q.w;
    double f = (sl < slthresh) ? sqrt((2 - 2 * wc) / sl) : acos(wc) / sqrt(sl);
    return mk3(2.0 * q.x * f, 2.0 * q.y * f, 2.0 * q.z * f);
}

__device__ __forceinline__ int robot_of(const DBatch& B, int vslot)
{
    return __builtin_amdgcn_readfirstlane(B.wave_robot[vslot >> 6]);
}

struct BondOut { d3 f1, m1, f2, m2; double strain1, strain2; bool diverged; };
// history of one bond (_LastPos2, _LastAngle1, _LastAngle2 + the small-angle flag); loaded by the caller BEFORE the
// arithmetic and stored after it, so that all memory operations of a bond are issued together (the kernels are
// latency-bound: 70% of wave time was s_waitcnt with loads scattered through the math)
struct BondHist { d3 pos2, ang1, ang2; bool small; bool store_hist, store_flag; };

__device__ __forceinline__ BondHist load_bond_hist(const DBatch& B, int slot)
{
    BondHist h;
    h.pos2 = mk3(HIST(0, slot), HIST(1, slot), HIST(2, slot));
    h.ang1 = mk3(HIST(3, slot), HIST(4, slot), HIST(5, slot));
    h.ang2 = mk3(HIST(6, slot), HIST(7, slot), HIST(8, slot));
    h.small = B.small_angle[slot] != 0;
    h.store_hist = h.store_flag = false;
    return h;
}
__device__ __forceinline__ void store_bond_hist(const DBatch& B, int slot, const BondHist& h)
{
    if (h.store_hist) {
        HIST(0, slot) = h.pos2.x; HIST(1, slot) = h.pos2.y; HIST(2, slot) = h.pos2.z;
        HIST(3, slot) = h.ang1.x; HIST(4, slot) = h.ang1.y; HIST(5, slot) = h.ang1.z;
        HIST(6, slot) = h.ang2.x; HIST(7, slot) = h.ang2.y; HIST(8, slot) = h.ang2.z;
    }
    if (h.store_flag) B.small_angle[slot] = h.small ? 1 : 0;
}

// One internal bond between voxel 1 (negative side) and voxel 2: pure arithmetic, `H` in/out.
__device__ __forceinline__ BondOut bond_compute(const DBatch& B, const DBondClass& C, int axis, BondHist& H,
                                                d3 p1, dq q1, double s1, d3 p2, dq q2, double s2,
                                                double dt_prev, double bond_z_half)
{
    BondOut o;
    const double nom_dist = (s1 + s2) * 0.5;
    d3 xrel = to_xdir(axis, p2 - p1);
    dq a1 = to_xdir(axis, q1), a2 = to_xdir(axis, q2);
    d3 rel = rotinv(a1, xrel);
    dq new2 = qmul(conj(a1), a2);

    // small/large-angle switch with hysteresis (VXS_BondInternal.cpp:72-77)
    bool small = H.small, changed = false;
    const double small_turn = (fabs(rel.z) + fabs(rel.y)) / rel.x;
    const double extend = rel.x / nom_dist;
    if (!small && new2.w > B.small_angle_w && small_turn < VXH_SA_BOND_BEND_RAD && extend < VXH_SA_BOND_EXT_PERC) { small = true; changed = true; }
    else if (small && (!(new2.w > B.smallish_angle_w) || small_turn > VXH_HYST * VXH_SA_BOND_BEND_RAD || extend > VXH_HYST * VXH_SA_BOND_EXT_PERC)) { small = false; changed = true; }
    H.small = small; H.store_flag = changed;

    d3 pos2, ang1, ang2;
    dq rot;
    if (small) {
        ang1 = mk3(0, 0, 0);
        ang2 = to_rotvec(new2, B.slthresh_acos2sqrt);
        pos2 = mk3(rel.x - nom_dist, rel.y, rel.z);
        rot = conj(a1);
    } else {
        dq align = from_angle_to_pos_x(rel);
        rot = qmul(align, conj(a1));
        pos2 = mk3(sqrt(len2(xrel)) - nom_dist, 0, 0);
        ang1 = to_rotvec(align, B.slthresh_acos2sqrt);
        ang2 = to_rotvec(qmul(rot, a2), B.slthresh_acos2sqrt);
    }

    // axial stress (UpdateBondStrain, VXS_BondInternal.cpp:189-307; linear materials)
    const double strain = pos2.x / C.L;
    double stress;
    o.strain1 = o.strain2 = strain;            // CurStrainV1 / CurStrainV2 (SetStrainDir), read by the land_water drag mesh
    if (C.homogeneous) stress = C.stress_E1 * strain;
    else {
        double e1 = strain, e2 = strain, t1 = C.stress_E1 * e1, t2 = C.stress_E2 * e2;
        double diff = fabs(t1 - t2), sum = fabs(t1 + t2);
        for (int it = 0; it < 3 && diff > sum * .0005; ++it) {
            e1 = 2 * t2 / (t1 + t2) * e1;
            e2 = 2 * t1 / (t1 + t2) * e2;
            t1 = C.stress_E1 * e1; t2 = C.stress_E2 * e2;
            diff = fabs(t1 - t2); sum = fabs(t1 + t2);
        }
        stress = (t1 + t2) / 2;
        o.strain1 = e1; o.strain2 = e2;
    }
    o.diverged = strain > 100;                 // VX_Sim.cpp:1775

    // beam equations (VXS_BondInternal.cpp:128-153)
    d3 f1 = mk3(stress * C.area_sum / 2, C.b1 * pos2.y - C.b2 * (ang1.z + ang2.z), C.b1 * pos2.z + C.b2 * (ang1.y + ang2.y));
    d3 f2 = -f1;
    d3 m1 = mk3(C.a2 * (ang1.x - ang2.x), C.b2 * pos2.z + C.b3 * (2 * ang1.y + ang2.y), -C.b2 * pos2.y + C.b3 * (2 * ang1.z + ang2.z));
    d3 m2 = mk3(C.a2 * (ang2.x - ang1.x), C.b2 * pos2.z + C.b3 * (ang1.y + 2 * ang2.y), -C.b2 * pos2.y + C.b3 * (ang1.z + 2 * ang2.z));

    // velocity damping from the finite-differenced bond-frame pose (AddDampForces :310-346); skipped on the step the
    // mode flips, and the history is only refreshed when it runs
    if (!changed) {
        if (dt_prev != 0) {
            const double inv = 1.0 / dt_prev;
            d3 v = (pos2 - H.pos2) * inv, w1 = (ang1 - H.ang1) * inv, w2 = (ang2 - H.ang2) * inv;
            const double z = bond_z_half;
            f1 = f1 + mk3(C.sq_a1m1 * v.x, C.sq_b1m1 * v.y - C.sq_b2fm1 * (w1.z + w2.z), C.sq_b1m1 * v.z + C.sq_b2fm1 * (w1.y + w2.y)) * z;
            if (!C.homogeneous)
                f2 = f2 + mk3(-C.sq_a1m2 * v.x, -C.sq_b1m2 * v.y + C.sq_b2fm2 * (w1.z + w2.z), -C.sq_b1m2 * v.z - C.sq_b2fm2 * (w1.y + w2.y)) * z;
            m1 = m1 + mk3(-C.sq_a2i1 * (w2.x - w1.x), C.sq_b2fm1 * v.z + C.sq_b3i1 * (2 * w1.y + w2.y), -C.sq_b2fm1 * v.y + C.sq_b3i1 * (2 * w1.z + w2.z)) * (0.5 * z);
            m2 = m2 + mk3(C.sq_a2i2 * (w2.x - w1.x), C.sq_b2fm2 * v.z + C.sq_b3i2 * (w1.y + 2 * w2.y), -C.sq_b2fm2 * v.y + C.sq_b3i2 * (w1.z + 2 * w2.z)) * (0.5 * z);
        }
        H.pos2 = pos2; H.ang1 = ang1; H.ang2 = ang2; H.store_hist = true;
    }

    // back to the global frame (:158-171)
    o.f1 = to_orig(axis, rotinv(rot, f1));
    o.f2 = C.homogeneous ? -o.f1 : to_orig(axis, rotinv(rot, f2));
    o.m1 = to_orig(axis, rotinv(rot, m1));
    o.m2 = to_orig(axis, rotinv(rot, m2));
    return o;
}

struct VoxState { d3 pos, lm, am; dq ang; double scale; };

// Everything of EulerStep after the internal-bond sums: collision bonds, gravity, floor, integration, actuation.
// F/M arrive holding slow damping + internal bond forces / minus internal bond moments.  Returns |new velocity|^2.
__device__ __forceinline__ double voxel_update(const DBatch& B, const DRobot& R, const DVoxClass& C, int v, int cur,
                                               double t, d3 F, d3 M, d3 vel, VoxState& S, int row, int ccnt, d3 drag)
{
    const int flags = R.flags;
    const double dt = R.dt;
    const bool fluid = (flags & RF_FLUID) != 0;
    if (ccnt > 0) {
        // collision bonds in creation order (VXS_Voxel.cpp:519-530, VXS_BondCollision.cpp:41-59); partner slots and the
        // pair stiffness a1 were stored by rebuild_rows.  Loads are issued four partners at a time: the loop is a
        // chain of dependent HBM/L2 accesses, not arithmetic.
        for (int k0 = 0; k0 < ccnt; k0 += 4) {
            int o[4]; double a1[4], qx[4], qy[4], qz[4], qs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = k0 + j < ccnt;
                const size_t at = (size_t)(k0 + j) * B.col_rows + row;   // partner-major: coalesced across the wave
                o[j] = on ? B.col_partner[at] : -1;
                a1[j] = on ? B.col_a1[at] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int oo = o[j] < 0 ? v : o[j];
                qx[j] = POS(cur, 0, oo); qy[j] = POS(cur, 1, oo); qz[j] = POS(cur, 2, oo); qs[j] = SCALE(cur, oo);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (o[j] < 0) continue;
                const bool second = o[j] < v;                          // Vox1 = the earlier surface voxel; am I Vox2?
                d3 d = mk3(qx[j] - S.pos.x, qy[j] - S.pos.y, qz[j] - S.pos.z);   // partner - me
                if (second) d = -d;                                    // Pos2 = pVox2 - pVox1
                const double nom = second ? (qs[j] + S.scale) * 0.75 : (S.scale + qs[j]) * 0.75;
                const double l = sqrt(len2(d));
                const double reld = nom - l;
                if (reld > 0) {
                    d3 f2 = ((d * (1.0 / l)) * a1[j]) * reld;          // force on Vox2
                    F = second ? F + f2 : F - f2;
                }
            }
        }
    }
    if ((flags & RF_GRAV) && !fluid) F.z += C.mass * R.grav_acc;
    if (fluid) F = F + drag;                  // LW/VXS_Voxel.cpp:372-373

    if ((flags & RF_FLOOR) && !fluid) {       // CalcFloorEffect, VXS_Voxel.cpp:708-758
        const double pen = 0.5 * S.scale - S.pos.z;
        bool static_fric = false;
        if (pen > 0) {
            const double normal = C.k_floor * pen;
            const double fz = normal - R.col_z * C.c_lin * vel.z;
            const double surf_vel = sqrt(vel.x * vel.x + vel.y * vel.y);
            const double surf_force = sqrt(F.x * F.x + F.y * F.y);
            const double fric = C.u_dynamic * normal;
            double fx = 0, fy = 0;
            bool stopped = (vel.x == 0 && vel.y == 0);
            if (flags & RF_STICKY) { S.lm.x = 0; S.lm.y = 0; static_fric = true; stopped = true; }
            if (stopped) {
                if (surf_force < C.u_static * normal) static_fric = true;
            } else if (fric * dt < C.mass * surf_vel) {
                // -(cos, sin)(atan2(vy, vx)) * fric == -(vx, vy)/|v| * fric
                const double inv = fric / surf_vel;
                fx = -vel.x * inv; fy = -vel.y * inv;
            } else { static_fric = true; S.lm.x = 0; S.lm.y = 0; }
            F.x += fx; F.y += fy; F.z += fz;
        }
        if (static_fric) { F.x = 0; F.y = 0; }
    }

    // EulerStep, VXS_Voxel.cpp:183-222
    S.lm = S.lm + F * dt;
    S.pos = S.pos + S.lm * (dt * C.mass_inv);
    S.am = S.am + M * dt;
    const double amf = 1 - 10 * R.slow_z * C.inertia_inv * C.c_ang * dt;
    S.am = S.am * amf;
    d3 w = S.am * C.inertia_inv;
    dq spin = qmul(mkq(0, w.x * 0.5, w.y * 0.5, w.z * 0.5), S.ang);
    dq ang = mkq(S.ang.w + spin.w * dt, S.ang.x + spin.x * dt, S.ang.y + spin.y * dt, S.ang.z + spin.z * dt);
    {
        const double l = sqrt(ang.x * ang.x + ang.y * ang.y + ang.z * ang.z + ang.w * ang.w);
        if (l != 0) { const double li = 1.0 / l; ang.w *= li; ang.x *= li; ang.y *= li; ang.z *= li; }
        if (ang.w >= 1.0) ang = mkq(1.0, 0, 0, 0);
    }
    S.ang = ang;

    // thermal actuation -> new scale (VXS_Voxel.cpp:224-340 without development; LW/VXS_Voxel.cpp:211-235)
    double new_scale;
    const double two_pi_f = (double)(2 * 3.1415926f);
    if (!(flags & RF_LW)) {
        const double c = (t >= 0.5 * R.init_cm_time) ? 1.0 : 2 * t / R.init_cm_time;
        const double prenatal = c * (((float)C.nom_size / C.nom_size) - 1);
        double ctrl = 0;
        if ((flags & RF_TEMP) && t >= R.init_cm_time)
            ctrl = (double)B.amp_damp[v] * ((double)R.temp_amplitude * sin(two_pi_f * (t / (double)R.temp_period + (double)B.phase[v]))) * C.cte;
        new_scale = ctrl * C.nom_size + (1 + prenatal) * C.nom_size;
        const double max_scale = (1 + R.growth_amplitude) * C.nom_size, min_scale = R.min_temp_fact * C.nom_size;
        if (new_scale < S.scale && new_scale < min_scale) new_scale = S.scale;
        if (new_scale > S.scale && new_scale > max_scale) new_scale = S.scale;
    } else {
        double tf = 1.0;
        if ((flags & RF_TEMP) && t >= R.init_cm_time)
            tf = 1 + ((double)R.temp_amplitude * sin(two_pi_f * (t / (double)R.temp_period + (double)B.phase[v]))) * C.cte;
        if (tf < 0.1) tf = 0.1;
        new_scale = tf * C.nom_size;
    }
    S.scale = new_scale;
    return len2(S.lm * C.mass_inv);
}

// ------------------------------------------------------------------------------------- per-robot step control
struct StepCtl { int go, latch, eol, rebuild; };

// Executed by ONE thread per robot before every step (and once after the last): completes the previous step's
// accounting, evaluates the stop condition and decides what this step needs.
__device__ __forceinline__ StepCtl step_control(const DRobot& R, DRobotState& rs, long long step_cap, int begin_new_step)
{
    StepCtl c; c.go = c.latch = c.eol = c.rebuild = 0;
    if (rs.status != 0) return c;
    if (rs.active) {                         // finish the step the previous round computed
        if (rs.diverged) rs.status = 2;
        else { rs.cur_time += R.dt; rs.steps += 1; rs.dt_prev = R.dt; }
        rs.active = 0;
    }
    if (rs.status != 0) return c;
    const double t = rs.cur_time;
    bool stop = false;                       // StopConditionMet, VX_Sim.cpp:1398-1423 (LW/VX_Sim.cpp:1160-1172)
    if ((R.flags & RF_LW) || !(t <= R.init_cm_time)) {
        if (R.stop_type == 1) stop = rs.steps > (int)(R.stop_value + 0.5);
        else if (R.stop_type == 2) stop = t > (R.stop_value + R.afterlife);
        else if (R.stop_type == 3) stop = t > R.temp_period_d * R.stop_value;
    }
    if (stop) { rs.status = 1; return c; }
    if (!begin_new_step || (long long)rs.steps >= step_cap) return c;
    c.go = 1;
    if (!rs.cm_init && t > R.init_cm_time) c.latch = 1;                          // VX_Sim.cpp:1064
    if (!(R.flags & RF_LW) && t >= R.stop_value && rs.eol_post_y == 0) c.eol = 1;   // :1078
    if (R.flags & RF_SELF_COL) {                                                 // UpdateCollisions :1729-1755
        const double mv = sqrt(__longlong_as_double((long long)rs.maxvel2_bits));
        rs.max_disp += fabs(mv * rs.dt_prev / R.lat);
        rs.maxvel2_bits = 0ull;
        if (!(R.flags & RF_HORIZON_COL) || rs.max_disp > (R.col_horizon - 1.0) / 2) { c.rebuild = 1; rs.max_disp = 0.0; rs.rebuilds += 1; }
    }
    rs.rebuild_now = c.rebuild;
    rs.active = 1;
    return c;
}

// IniCM latch (= SS.CurCM of the previous step: mass-weighted SEQUENTIAL sum in voxel order, GetCM VX_Sim.cpp:2415-2430)
// and EndOfLifetimePosteriorY (getPosteriorY :2640-2656).  Whole workgroup; `sh` holds 4*CH doubles of LDS scratch.
__device__ __forceinline__ void latch_cm(const DBatch& B, const DRobot& R, DRobotState& rs, int cur, bool latch, bool eol, double* sh, int CH)
{
    const int tid = threadIdx.x, T = blockDim.x, base = R.vox_begin;
    double sx = 0, sy = 0, sz = 0, sm = 0, miny = 100000.0;
    for (int c0 = 0; c0 < R.nvox; c0 += CH) {
        for (int k = tid; k < CH && c0 + k < R.nvox; k += T) {
            const int g = base + c0 + k;
            const DVoxClass& C = B.vclass_tab[B.vclass[g]];
            sh[k] = POS(cur, 0, g); sh[CH + k] = POS(cur, 1, g); sh[2 * CH + k] = POS(cur, 2, g);
            sh[3 * CH + k] = (C.mat == 5) ? -C.mass : C.mass;   // sign marks the material excluded from PosteriorY
        }
        __syncthreads();
        if (tid == 0) {
            const int n = min(CH, R.nvox - c0);
            for (int k = 0; k < n; ++k) {
                const double m = fabs(sh[3 * CH + k]);
                sx = __dadd_rn(sx, __dmul_rn(sh[k], m)); sy = __dadd_rn(sy, __dmul_rn(sh[CH + k], m)); sz = __dadd_rn(sz, __dmul_rn(sh[2 * CH + k], m)); sm += m;
                if (!(sh[3 * CH + k] < 0)) { const double y = sh[CH + k] / R.lat; if (y < miny) miny = y; }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (latch) { const double inv = 1.0 / sm; rs.ini_cm[0] = inv * sx; rs.ini_cm[1] = inv * sy; rs.ini_cm[2] = inv * sz; rs.cm_init = 1; }
        if (eol) rs.eol_post_y = miny;
    }
}

// CalcL1Bonds (VX_Sim.cpp:2357-2413).  The calling workgroup builds the partner rows of surface voxels
// [i_begin, i_begin + blockDim.x) of robot R: every surface voxel tests all others (staged through LDS in chunks of
// CH) and keeps, in ascending partner order = creation order of its collision bonds in the reference, those that
// pass the distance filter, are more than `hops` bonds away and lie within CollisionHorizon scaled voxel sizes.
// `sh`: 4*CH doubles + CH ints of LDS.
__device__ __forceinline__ void rebuild_rows(const DBatch& B, const DRobot& R, DRobotState& rs, int cur, int i_begin, double* sh, int CH)
{
    const int tid = threadIdx.x, T = blockDim.x;
    int* shv = (int*)(sh + 4 * CH);
    const int i = i_begin + tid;
    const bool mine = i < R.nsurf;
    int vi = 0, cnt = 0;
    const unsigned long long* row = B.excl + R.excl_begin + (long long)(mine ? i : 0) * R.excl_wpr;
    d3 pi = mk3(0, 0, 0); double si = 0;
    if (mine) {
        vi = B.surf[R.surf_begin + i];
        pi = mk3(POS(cur, 0, vi), POS(cur, 1, vi), POS(cur, 2, vi));
        si = SCALE(cur, vi);
    }
    const double H = R.col_horizon;
    for (int c0 = 0; c0 < R.nsurf; c0 += CH) {
        const int n = min(CH, R.nsurf - c0);
        for (int k = tid; k < n; k += T) {
            const int vj = B.surf[R.surf_begin + c0 + k];
            sh[k] = POS(cur, 0, vj); sh[CH + k] = POS(cur, 1, vj); sh[2 * CH + k] = POS(cur, 2, vj);
            sh[3 * CH + k] = SCALE(cur, vj); shv[k] = vj;
        }
        __syncthreads();
        if (mine) {
            unsigned long long word = 0;
            for (int k = 0; k < n; ++k) {
                const int j = c0 + k;
                if ((j & 63) == 0) word = row[j >> 6];            // chunk starts are multiples of 64
                if (j == i) continue;
                const d3 d = pi - mk3(sh[k], sh[CH + k], sh[2 * CH + k]);
                const double d2 = len2(d);
                if (!(d2 < R.filter_dist2)) continue;
                if ((word >> (j & 63)) & 1ull) continue;          // !pV1->IsNearbyVox(SIndex2)
                const double s1 = (j > i) ? si : sh[3 * CH + k];   // scale of Vox1 = the earlier one, used twice (:2382)
                const double act = H * (s1 + s1) * 0.5;
                if (d2 < act * act) {
                    if (cnt < VXH_MAXCOL) {
                        const int vj = shv[k];
                        const DVoxClass& Ci = B.vclass_tab[B.vclass[vi]];   // CVX_Bond::LinkVoxels + UpdateConstants for the pair
                        const DVoxClass& Cj = B.vclass_tab[B.vclass[vj]];
                        const double E1 = (j > i) ? Ci.E : Cj.E, E2 = (j > i) ? Cj.E : Ci.E;
                        const double E = (E1 * E2 / (E1 + E2)) * 2;
                        const double L = (Ci.nom_size + Cj.nom_size) * 0.5;
                        const size_t at = (size_t)cnt * B.col_rows + (R.surf_begin + i);
                        B.col_partner[at] = vj;
                        B.col_a1[at] = E * (L * L) / L;
                    }
                    ++cnt;
                }
            }
        }
        __syncthreads();
    }
    if (mine) {
        if (cnt > VXH_MAXCOL) { cnt = VXH_MAXCOL; atomicOr(&rs.col_overflow, 1); }
        B.col_cnt[R.surf_begin + i] = cnt;
    }
}

// land_water fluid drag (LW/VX_Sim.cpp:1516-1597).  Phase 1: every deformable surface vertex = mean over the <= 7 voxels
// touching that lattice corner of Pos + R(Angle) * corner offset, corner offsets from the bond strains of the PREVIOUS step
// (CornerPosCur/CornerNegCur, LW/VXS_Voxel.cpp:472-475; GetCurVLoc LW/VX_MeshUtil.cpp:388-428) -> LDS.  Phase 2: every voxel
// sums the quadratic drag of the two triangles on each of its exposed faces, in the reference's facet order.
__device__ __forceinline__ d3 rot_fwd(dq q, d3 f)      // CQuat::RotateVec3D, Vec3D.h:293-299
{
    double tw = f.x * q.x + f.y * q.y + f.z * q.z;
    double tx = f.x * q.w - f.y * q.z + f.z * q.y;
    double ty = f.x * q.z + f.y * q.w - f.z * q.x;
    double tz = -f.x * q.y + f.y * q.x + f.z * q.w;
    return mk3(q.w * tx + q.x * tw + q.y * tz - q.z * ty, q.w * ty - q.x * tz + q.y * tw + q.z * tx, q.w * tz + q.x * ty - q.y * tx + q.z * tw);
}
__device__ __forceinline__ d3 cross3(d3 a, d3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ double dot3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 normalized3(d3 a) { const double l = sqrt(len2(a)); return l > 0 ? a * (1.0 / l) : a; }

template <int BLOCK>
__device__ __forceinline__ void fluid_drag(const DBatch& B, const DRobot& R, int cur, double* sh, bool valid, int v, double mass_inv, double nom)
{
    const unsigned nv = B.nv, tm = B.total_mv;
    for (int i = threadIdx.x; i < R.nmv; i += BLOCK) {
        const int gi = R.vert_begin + i;
        d3 avg = mk3(0, 0, 0); double tw = 0;
        for (int q = 0; q < 8; ++q) {
            const int comp = B.vert_comp[(unsigned)q * tm + gi];
            if (comp < 0) break;
            const int u = comp >> 3, corner = comp & 7;
            const d3 cp = mk3((1 + B.strain[u]) * nom * 0.5, (1 + B.strain[nv + u]) * nom * 0.5, (1 + B.strain[2 * nv + u]) * nom * 0.5);
            const d3 cn = mk3(-(1 + B.strain[3 * nv + u]) * nom * 0.5, -(1 + B.strain[4 * nv + u]) * nom * 0.5, -(1 + B.strain[5 * nv + u]) * nom * 0.5);
            const d3 off = mk3((corner & 4) ? cp.x : cn.x, (corner & 2) ? cp.y : cn.y, (corner & 1) ? cp.z : cn.z);
            const d3 p = mk3(POS(cur, 0, u), POS(cur, 1, u), POS(cur, 2, u)) + rot_fwd(mkq(QUAT(0, u), QUAT(1, u), QUAT(2, u), QUAT(3, u)), off);
            avg = avg + p; tw += 1.0;
        }
        const double inv = 1.0 / tw;
        const d3 v0 = mk3(B.vert_v0[gi], B.vert_v0[tm + gi], B.vert_v0[2 * tm + gi]);
        const d3 np = avg * inv;
        const d3 now = v0 + (np - v0);                               // v + DrawOffset, as the reference stores it
        sh[i] = now.x; sh[R.nmv + i] = now.y; sh[2 * R.nmv + i] = now.z;
    }
    __syncthreads();
    if (valid) {
        d3 drag = mk3(0, 0, 0);
        const unsigned mask = B.open_face[v];
        if (mask) {
            const d3 speed = mk3(LINMOM(0, v), LINMOM(1, v), LINMOM(2, v)) * mass_inv;
            const d3 sdir = normalized3(speed);
            // corner codes (NNN..PPP) of the two triangles of faces +X,-X,+Y,-Y,+Z,-Z (LW/VX_MeshUtil.cpp:165-189)
            const unsigned tri[6][2] = {{0x467u, 0x475u}, {0x032u, 0x013u}, {0x237u, 0x276u}, {0x051u, 0x045u}, {0x157u, 0x173u}, {0x064u, 0x026u}};
            for (int d = 0; d < 6; ++d) {
                if (!(mask & (1u << d))) continue;
                for (int t = 0; t < 2; ++t) {
                    const unsigned code = tri[d][t];
                    const int ia = B.corner_vert[((code >> 8) & 7u) * nv + v], ib = B.corner_vert[((code >> 4) & 7u) * nv + v], ic = B.corner_vert[(code & 7u) * nv + v];
                    const d3 A = mk3(sh[ia], sh[R.nmv + ia], sh[2 * R.nmv + ia]);
                    const d3 AB = mk3(sh[ib], sh[R.nmv + ib], sh[2 * R.nmv + ib]) - A, AC = mk3(sh[ic], sh[R.nmv + ic], sh[2 * R.nmv + ic]) - A;
                    const d3 cr = cross3(AB, AC);
                    const double area = fabs(sqrt(len2(cr)) / 2.0);
                    const d3 n = normalized3(cr);                       // CalcFaceNormals
                    const float ang = (float)acos(dot3(sdir, normalized3(n)));
                    if (fabsf(ang) < VXH_PI / 2) {
                        const d3 proj = normalized3(n) * dot3(speed, n);    // ProjectOnTo
                        drag = drag + normalized3(proj) * (-R.drag_coef * area * len2(proj));
                    }
                }
            }
        }
        B.dragf[v] = drag.x; B.dragf[nv + v] = drag.y; B.dragf[2 * nv + v] = drag.z;
    }
}

// ================================================================================================ fused path
// LDS: exchange buffer ex[axis][component 0..5][BLOCK] doubles (Force2, Moment2 of the bond whose POSITIVE end is
// voxel `local`), reused as scratch by latch_cm / rebuild_rows.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK, BLOCK / 256) void k_robot_steps(DBatch B, const int* __restrict__ robot_list, long long step_cap, int iters)
{
    extern __shared__ __align__(16) double ex[];
    __shared__ int s_go, s_latch, s_eol, s_rebuild, s_cur, s_div;
    __shared__ double s_time, s_dtprev;
    const int r = robot_list[blockIdx.x];   // robots of one size class, longest-running first
    const DRobot& R = B.robot[r];
    // the robot's mutable control block lives in LDS for the whole launch: the per-step control is a serial chain
    // of ~20 dependent accesses executed by one thread while the workgroup waits, so it must not touch HBM
    __shared__ DRobotState rs;
    const int tid = threadIdx.x;
    if (tid == 0) rs = B.rstate[r];
    const bool valid = tid < R.nvox;
    const int v = R.vox_begin + tid;
    const int nv = B.nv;

    // class-constant tables into LDS (a handful of entries for a whole population): per-lane gathers of ~20 doubles
    // per bond then cost LDS reads instead of vector-memory loads
    __shared__ DBondClass s_bct[VXH_LDS_BCLASS];
    __shared__ DVoxClass s_vct[VXH_LDS_VCLASS];
    const bool tabs_in_lds = B.n_bclass <= VXH_LDS_BCLASS && B.n_vclass <= VXH_LDS_VCLASS;
    if (tabs_in_lds) {
        for (int k = tid; k < B.n_bclass * (int)(sizeof(DBondClass) / 8); k += BLOCK) ((double*)s_bct)[k] = ((const double*)B.bclass_tab)[k];
        for (int k = tid; k < B.n_vclass * (int)(sizeof(DVoxClass) / 8); k += BLOCK) ((double*)s_vct)[k] = ((const double*)B.vclass_tab)[k];
    }
    __syncthreads();
    const DBondClass* bct = tabs_in_lds ? s_bct : B.bclass_tab;
    const DVoxClass& C = (tabs_in_lds ? (const DVoxClass*)s_vct : B.vclass_tab)[valid ? B.vclass[v] : 0];
    int row = -1;                              // my row of collision partners (surface voxels of colliding robots)
    if (valid && (R.flags & RF_SELF_COL)) { const int so = B.surf_ord[v]; if (so >= 0) row = R.surf_begin + so; }

    for (int it = 0; it <= iters; ++it) {
        if (tid == 0) {
            StepCtl c = step_control(R, rs, step_cap, it < iters);
            s_go = c.go; s_latch = c.latch; s_eol = c.eol; s_rebuild = c.rebuild;
            s_cur = rs.steps & 1; s_time = rs.cur_time; s_dtprev = rs.dt_prev; s_div = 0;
        }
        __syncthreads();
        if (!s_go) break;
        const int cur = s_cur, nxt = cur ^ 1;
        if (s_latch || s_eol) latch_cm(B, R, rs, cur, s_latch != 0, s_eol != 0, ex, BLOCK);
        if (s_rebuild) { for (int i0 = 0; i0 < R.nsurf; i0 += BLOCK) rebuild_rows(B, R, rs, cur, i0, ex, 2 * BLOCK); __syncthreads(); }
        const int ccnt = (row >= 0 && !(B.dbg & 1)) ? B.col_cnt[row] : 0;   // issued early, consumed in the voxel phase
        if (R.flags & RF_FLUID) { fluid_drag<BLOCK>(B, R, cur, ex, valid, v, C.mass_inv, R.lat); __syncthreads(); }

        // ---- bond phase: this voxel's +X, +Y, +Z bonds; own-side sums stay in registers, far-side outputs go to LDS
        d3 F = mk3(0, 0, 0), M = mk3(0, 0, 0);
        if (valid) {
            int vb = v;                        // opaque copy: keeps address arithmetic out of the step loop's live ranges
            asm volatile("" : "+v"(vb));
            const d3 p1 = mk3(POS(cur, 0, vb), POS(cur, 1, vb), POS(cur, 2, vb));
            const dq q1 = mkq(QUAT(0, vb), QUAT(1, vb), QUAT(2, vb), QUAT(3, vb));
            const double sc1 = SCALE(cur, vb);
            bool div = false;
#pragma unroll 1
            for (int a = 0; a < 3; ++a) {
                const int bc = B.bclass[a * nv + vb];
                if (bc < 0) continue;
                const int v2 = B.nbr[(2 * a) * nv + vb];
                const d3 p2 = mk3(POS(cur, 0, v2), POS(cur, 1, v2), POS(cur, 2, v2));
                const dq q2 = mkq(QUAT(0, v2), QUAT(1, v2), QUAT(2, v2), QUAT(3, v2));
                const double sc2 = SCALE(cur, v2);
                BondHist H = load_bond_hist(B, a * nv + vb);
                BondOut o = bond_compute(B, bct[bc], a, H, p1, q1, sc1, p2, q2, sc2, s_dtprev, R.bond_z_half);
                store_bond_hist(B, a * nv + vb, H);
                F = F + o.f1; M = M - o.m1;
                div = div || o.diverged;
                if (R.flags & RF_FLUID) {     // SetStrainDir (VXS_BondInternal.cpp:300-304): my +a side, the neighbour's -a side
                    B.strain[(unsigned)a * nv + vb] = o.strain1;
                    B.strain[(unsigned)(3 + a) * nv + v2] = o.strain2;
                }
                double* e = ex + (a * 6) * BLOCK + (v2 - R.vox_begin);
                e[0] = o.f2.x; e[BLOCK] = o.f2.y; e[2 * BLOCK] = o.f2.z; e[3 * BLOCK] = o.m2.x; e[4 * BLOCK] = o.m2.y; e[5 * BLOCK] = o.m2.z;
            }
            if (div) s_div = 1;
        }
        __syncthreads();
        if (s_div) {                           // Integrate() returns before the voxel loop (VX_Sim.cpp:1777)
            if (tid == 0) rs.diverged = 1;
            continue;                          // next step_control marks the robot diverged
        }
        // ---- voxel phase
        double vel2 = 0;
        if (valid) {
            int vx = v;
            asm volatile("" : "+v"(vx));
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (B.nbr[(2 * a + 1) * nv + vx] < 0) continue;
                const double* e = ex + (a * 6) * BLOCK + tid;
                F = F + mk3(e[0], e[BLOCK], e[2 * BLOCK]);
                M = M - mk3(e[3 * BLOCK], e[4 * BLOCK], e[5 * BLOCK]);
            }
            VoxState S;
            S.pos = mk3(POS(cur, 0, vx), POS(cur, 1, vx), POS(cur, 2, vx));
            S.ang = mkq(QUAT(0, vx), QUAT(1, vx), QUAT(2, vx), QUAT(3, vx));
            S.scale = SCALE(cur, vx);
            S.lm = mk3(LINMOM(0, vx), LINMOM(1, vx), LINMOM(2, vx));
            S.am = mk3(ANGMOM(0, vx), ANGMOM(1, vx), ANGMOM(2, vx));
            const d3 vel = S.lm * C.mass_inv;
            F = F + (vel * (-R.slow_z)) * C.c_lin;
            d3 drag = mk3(0, 0, 0);
            if (R.flags & RF_FLUID) drag = mk3(B.dragf[vx], B.dragf[(unsigned)nv + vx], B.dragf[2u * nv + vx]);
            vel2 = voxel_update(B, R, C, vx, cur, s_time, F, M, vel, S, row, ccnt, drag);
            POS(nxt, 0, vx) = S.pos.x; POS(nxt, 1, vx) = S.pos.y; POS(nxt, 2, vx) = S.pos.z;
            SCALE(nxt, vx) = S.scale;
            LINMOM(0, vx) = S.lm.x; LINMOM(1, vx) = S.lm.y; LINMOM(2, vx) = S.lm.z;
            ANGMOM(0, vx) = S.am.x; ANGMOM(1, vx) = S.am.y; ANGMOM(2, vx) = S.am.z;
            QUAT(0, vx) = S.ang.w; QUAT(1, vx) = S.ang.x; QUAT(2, vx) = S.ang.y; QUAT(3, vx) = S.ang.z;
        }
        if ((R.flags & RF_SELF_COL) && !(B.dbg & 2)) {            // SS.MaxVoxVel for the collision horizon (VX_Sim.cpp:1625-1649)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { double o2 = __shfl_xor(vel2, off); vel2 = o2 > vel2 ? o2 : vel2; }
            if ((tid & 63) == 0) atomicMax(&rs.maxvel2_bits, (unsigned long long)__double_as_longlong(vel2));
        }
        __syncthreads();                       // new poses visible to the whole workgroup, ex[] free again
    }
    if (tid == 0) B.rstate[r] = rs;
}

// ============================================================================================ streaming path
__global__ __launch_bounds__(256) void k_step_begin(DBatch B, long long step_cap, int begin_new_step)
{
    const int r = blockIdx.x;
    const DRobot& R = B.robot[r];
    DRobotState& rs = B.rstate[r];
    __shared__ double sh[4 * 256];
    __shared__ int s_go, s_latch, s_eol;
    if (threadIdx.x == 0) {
        StepCtl c = step_control(R, rs, step_cap, begin_new_step);
        s_go = c.go; s_latch = c.latch; s_eol = c.eol;
    }
    __syncthreads();
    if (!s_go) return;
    if (s_latch || s_eol) latch_cm(B, R, rs, rs.steps & 1, s_latch != 0, s_eol != 0, sh, 256);
}

// blocks [0, bond_blocks): one thread per bond slot; blocks beyond: collision-list rebuilds (reb_robot/reb_i0 tables),
// which overlap with the bond work of the other robots
__global__ __launch_bounds__(256) void k_bonds(DBatch B, int bond_blocks, const int* __restrict__ reb_robot, const int* __restrict__ reb_i0)
{
    if ((int)blockIdx.x >= bond_blocks) {
        __shared__ double sh[4 * 512 + 256];
        const int k = blockIdx.x - bond_blocks;
        const int r = reb_robot[k];
        DRobotState& rs = B.rstate[r];
        if (!rs.active || !rs.rebuild_now) return;
        rebuild_rows(B, B.robot[r], rs, rs.steps & 1, reb_i0[k], sh, 512);
        return;
    }
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= 3 * B.nv) return;
    const int axis = tid / B.nv;            // wave-uniform: nv is a multiple of 64
    const int v1 = tid - axis * B.nv;
    const int r = robot_of(B, v1);
    if (r < 0) return;
    const DRobotState& rs = B.rstate[r];
    if (!rs.active) return;
    const int bc = B.bclass[tid];
    if (bc < 0) return;
    const int v2 = B.nbr[(2 * axis) * B.nv + v1];
    const DRobot& R = B.robot[r];
    const int cur = rs.steps & 1;
    d3 p1 = mk3(POS(cur, 0, v1), POS(cur, 1, v1), POS(cur, 2, v1));
    d3 p2 = mk3(POS(cur, 0, v2), POS(cur, 1, v2), POS(cur, 2, v2));
    dq q1 = mkq(QUAT(0, v1), QUAT(1, v1), QUAT(2, v1), QUAT(3, v1));
    dq q2 = mkq(QUAT(0, v2), QUAT(1, v2), QUAT(2, v2), QUAT(3, v2));
    const double sc1 = SCALE(cur, v1), sc2 = SCALE(cur, v2);
    BondHist H = load_bond_hist(B, tid);
    BondOut o = bond_compute(B, B.bclass_tab[bc], axis, H, p1, q1, sc1, p2, q2, sc2, rs.dt_prev, R.bond_z_half);
    store_bond_hist(B, tid, H);
    if (o.diverged) atomicOr(&B.rstate[r].diverged, 1);
    BOUT(0, tid) = o.f1.x; BOUT(1, tid) = o.f1.y; BOUT(2, tid) = o.f1.z;
    BOUT(3, tid) = o.m1.x; BOUT(4, tid) = o.m1.y; BOUT(5, tid) = o.m1.z;
    BOUT(6, tid) = o.f2.x; BOUT(7, tid) = o.f2.y; BOUT(8, tid) = o.f2.z;
    BOUT(9, tid) = o.m2.x; BOUT(10, tid) = o.m2.y; BOUT(11, tid) = o.m2.z;
}

__global__ __launch_bounds__(256) void k_voxels(DBatch B)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= B.nv) return;
    const int r = robot_of(B, v);
    if (r < 0) return;
    DRobotState& rs = B.rstate[r];
    if (!rs.active || rs.diverged) return;    // Integrate() returns before the voxel loop when a bond diverged
    const DRobot& R = B.robot[r];
    const bool valid = (v - R.vox_begin) < R.nvox;   // padding slots stay in the wave for the reduction below
    double vel2 = 0;
    if (valid) {
        const DVoxClass& C = B.vclass_tab[B.vclass[v]];
        const int cur = rs.steps & 1, nxt = cur ^ 1;
        VoxState S;
        S.pos = mk3(POS(cur, 0, v), POS(cur, 1, v), POS(cur, 2, v));
        S.lm = mk3(LINMOM(0, v), LINMOM(1, v), LINMOM(2, v));
        S.am = mk3(ANGMOM(0, v), ANGMOM(1, v), ANGMOM(2, v));
        S.ang = mkq(QUAT(0, v), QUAT(1, v), QUAT(2, v), QUAT(3, v));
        S.scale = SCALE(cur, v);
        d3 vel = S.lm * C.mass_inv;           // Vel as left by the previous EulerStep (VXS_Voxel.cpp:409)
        // CalcTotalForce / CalcTotalMoment: fixed order PX,NX,PY,NY,PZ,NZ (VXS_Voxel.cpp:496-501,659-665)
        d3 F = (vel * (-R.slow_z)) * C.c_lin;
        d3 M = mk3(0, 0, 0);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (B.nbr[(2 * a) * B.nv + v] >= 0) {
                const int s = a * B.nv + v;
                F = F + mk3(BOUT(0, s), BOUT(1, s), BOUT(2, s));
                M = M - mk3(BOUT(3, s), BOUT(4, s), BOUT(5, s));
            }
            const int n = B.nbr[(2 * a + 1) * B.nv + v];
            if (n >= 0) {
                const int s = a * B.nv + n;
                F = F + mk3(BOUT(6, s), BOUT(7, s), BOUT(8, s));
                M = M - mk3(BOUT(9, s), BOUT(10, s), BOUT(11, s));
            }
        }
        int row = -1, ccnt = 0;
        if (R.flags & RF_SELF_COL) { const int so = B.surf_ord[v]; if (so >= 0) { row = R.surf_begin + so; ccnt = B.col_cnt[row]; } }
        vel2 = voxel_update(B, R, C, v, cur, rs.cur_time, F, M, vel, S, row, ccnt, mk3(0, 0, 0));
        POS(nxt, 0, v) = S.pos.x; POS(nxt, 1, v) = S.pos.y; POS(nxt, 2, v) = S.pos.z;
        SCALE(nxt, v) = S.scale;
        LINMOM(0, v) = S.lm.x; LINMOM(1, v) = S.lm.y; LINMOM(2, v) = S.lm.z;
        ANGMOM(0, v) = S.am.x; ANGMOM(1, v) = S.am.y; ANGMOM(2, v) = S.am.z;
        QUAT(0, v) = S.ang.w; QUAT(1, v) = S.ang.x; QUAT(2, v) = S.ang.y; QUAT(3, v) = S.ang.z;
    }
    if (R.flags & RF_SELF_COL) {              // SS.MaxVoxVel for the collision horizon (VX_Sim.cpp:1625-1649)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { double o = __shfl_xor(vel2, off); vel2 = o > vel2 ? o : vel2; }
        if ((threadIdx.x & 63) == 0) atomicMax(&rs.maxvel2_bits, (unsigned long long)__double_as_longlong(vel2));
    }
}

}  // namespace vxh
