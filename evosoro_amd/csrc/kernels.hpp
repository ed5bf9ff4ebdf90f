// HIP kernels of the batched Voxelyze time-stepper (gfx950).  FP64 throughout, one thread per bond slot /
// per voxel, SoA state in HBM (device_types.hpp).  Reference loops being replaced:
//   k_step_begin   CVX_Sim::TimeStep prologue + StopConditionMet + UpdateCollisions/CalcL1Bonds
//                  (VX_Sim.cpp:1054-1135,1398-1423,1729-1755,2357-2413) and the `CurTime += dt` epilogue (:1929)
//   k_bonds        the bond loop of CVX_Sim::Integrate (:1773-1776) = CVXS_BondInternal::CalcLinForce,
//                  UpdateBondStrain, AddDampForces (VXS_BondInternal.cpp:56-346)
//   k_voxels       the voxel loop (:1913) = CVXS_Voxel::EulerStep, CalcTotalForce, CalcTotalMoment,
//                  CalcFloorEffect (VXS_Voxel.cpp:169-758) + CVXS_BondCollision::CalcContactForce
//                  (VXS_BondCollision.cpp:41-59) + the MaxVoxVel part of UpdateStats (:1625-1649)
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.hpp"

namespace vxh {

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

// ------------------------------------------------------------------------------------------------ bonds
__global__ __launch_bounds__(256) void k_bonds(DBatch B)
{
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
    const DBondClass& C = B.bclass_tab[bc];
    const int cur = rs.steps & 1;

    d3 p1 = mk3(B.pos[cur][0][v1], B.pos[cur][1][v1], B.pos[cur][2][v1]);
    d3 p2 = mk3(B.pos[cur][0][v2], B.pos[cur][1][v2], B.pos[cur][2][v2]);
    dq q1 = mkq(B.quat[0][v1], B.quat[1][v1], B.quat[2][v1], B.quat[3][v1]);
    dq q2 = mkq(B.quat[0][v2], B.quat[1][v2], B.quat[2][v2], B.quat[3][v2]);
    const double nom_dist = (B.scale[cur][v1] + B.scale[cur][v2]) * 0.5;

    d3 xrel = to_xdir(axis, p2 - p1);
    dq a1 = to_xdir(axis, q1), a2 = to_xdir(axis, q2);
    d3 rel = rotinv(a1, xrel);
    dq new2 = qmul(conj(a1), a2);

    // small/large-angle switch with hysteresis (VXS_BondInternal.cpp:72-77)
    bool small = B.small_angle[tid] != 0, changed = false;
    const double small_turn = (fabs(rel.z) + fabs(rel.y)) / rel.x;
    const double extend = rel.x / nom_dist;
    if (!small && new2.w > B.small_angle_w && small_turn < VXH_SA_BOND_BEND_RAD && extend < VXH_SA_BOND_EXT_PERC) { small = true; changed = true; }
    else if (small && (!(new2.w > B.smallish_angle_w) || small_turn > VXH_HYST * VXH_SA_BOND_BEND_RAD || extend > VXH_HYST * VXH_SA_BOND_EXT_PERC)) { small = false; changed = true; }
    if (changed) B.small_angle[tid] = small ? 1 : 0;

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
    if (C.homogeneous) stress = C.stress_E1 * strain;
    else {
        double e1 = strain, e2 = strain, s1 = C.stress_E1 * e1, s2 = C.stress_E2 * e2;
        double diff = fabs(s1 - s2), sum = fabs(s1 + s2);
        for (int it = 0; it < 3 && diff > sum * .0005; ++it) {
            e1 = 2 * s2 / (s1 + s2) * e1;
            e2 = 2 * s1 / (s1 + s2) * e2;
            s1 = C.stress_E1 * e1; s2 = C.stress_E2 * e2;
            diff = fabs(s1 - s2); sum = fabs(s1 + s2);
        }
        stress = (s1 + s2) / 2;
    }
    if (strain > 100) atomicOr(&B.rstate[r].diverged, 1);   // VX_Sim.cpp:1775

    // beam equations (VXS_BondInternal.cpp:128-153)
    d3 f1 = mk3(stress * C.area_sum / 2, C.b1 * pos2.y - C.b2 * (ang1.z + ang2.z), C.b1 * pos2.z + C.b2 * (ang1.y + ang2.y));
    d3 f2 = -f1;
    d3 m1 = mk3(C.a2 * (ang1.x - ang2.x), C.b2 * pos2.z + C.b3 * (2 * ang1.y + ang2.y), -C.b2 * pos2.y + C.b3 * (2 * ang1.z + ang2.z));
    d3 m2 = mk3(C.a2 * (ang2.x - ang1.x), C.b2 * pos2.z + C.b3 * (ang1.y + 2 * ang2.y), -C.b2 * pos2.y + C.b3 * (ang1.z + 2 * ang2.z));

    // velocity damping from finite-differenced bond-frame pose (AddDampForces :310-346); skipped on the step the
    // mode flips, and the history is only refreshed when it runs
    if (!changed) {
        const double dtp = rs.dt_prev;
        if (dtp != 0) {
            const double inv = 1.0 / dtp;
            d3 v = mk3((pos2.x - B.hist[0][tid]) * inv, (pos2.y - B.hist[1][tid]) * inv, (pos2.z - B.hist[2][tid]) * inv);
            d3 w1 = mk3((ang1.x - B.hist[3][tid]) * inv, (ang1.y - B.hist[4][tid]) * inv, (ang1.z - B.hist[5][tid]) * inv);
            d3 w2 = mk3((ang2.x - B.hist[6][tid]) * inv, (ang2.y - B.hist[7][tid]) * inv, (ang2.z - B.hist[8][tid]) * inv);
            const double z = R.bond_z_half;
            f1 = f1 + mk3(C.sq_a1m1 * v.x, C.sq_b1m1 * v.y - C.sq_b2fm1 * (w1.z + w2.z), C.sq_b1m1 * v.z + C.sq_b2fm1 * (w1.y + w2.y)) * z;
            if (!C.homogeneous)
                f2 = f2 + mk3(-C.sq_a1m2 * v.x, -C.sq_b1m2 * v.y + C.sq_b2fm2 * (w1.z + w2.z), -C.sq_b1m2 * v.z - C.sq_b2fm2 * (w1.y + w2.y)) * z;
            m1 = m1 + mk3(-C.sq_a2i1 * (w2.x - w1.x), C.sq_b2fm1 * v.z + C.sq_b3i1 * (2 * w1.y + w2.y), -C.sq_b2fm1 * v.y + C.sq_b3i1 * (2 * w1.z + w2.z)) * (0.5 * z);
            m2 = m2 + mk3(C.sq_a2i2 * (w2.x - w1.x), C.sq_b2fm2 * v.z + C.sq_b3i2 * (w1.y + 2 * w2.y), -C.sq_b2fm2 * v.y + C.sq_b3i2 * (w1.z + 2 * w2.z)) * (0.5 * z);
        }
        B.hist[0][tid] = pos2.x; B.hist[1][tid] = pos2.y; B.hist[2][tid] = pos2.z;
        B.hist[3][tid] = ang1.x; B.hist[4][tid] = ang1.y; B.hist[5][tid] = ang1.z;
        B.hist[6][tid] = ang2.x; B.hist[7][tid] = ang2.y; B.hist[8][tid] = ang2.z;
    }

    // back to the global frame (:158-171)
    f1 = to_orig(axis, rotinv(rot, f1));
    f2 = C.homogeneous ? -f1 : to_orig(axis, rotinv(rot, f2));
    m1 = to_orig(axis, rotinv(rot, m1));
    m2 = to_orig(axis, rotinv(rot, m2));
    B.bout[0][tid] = f1.x; B.bout[1][tid] = f1.y; B.bout[2][tid] = f1.z;
    B.bout[3][tid] = m1.x; B.bout[4][tid] = m1.y; B.bout[5][tid] = m1.z;
    B.bout[6][tid] = f2.x; B.bout[7][tid] = f2.y; B.bout[8][tid] = f2.z;
    B.bout[9][tid] = m2.x; B.bout[10][tid] = m2.y; B.bout[11][tid] = m2.z;
}

// ------------------------------------------------------------------------------------------------ voxels
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
    double vel2_new = 0;
    if (valid) {
    const DVoxClass& C = B.vclass_tab[B.vclass[v]];
    const int cur = rs.steps & 1, nxt = cur ^ 1;
    const double dt = R.dt;
    const int flags = R.flags;
    const bool fluid = (flags & RF_FLUID) != 0;

    d3 pos = mk3(B.pos[cur][0][v], B.pos[cur][1][v], B.pos[cur][2][v]);
    d3 lm = mk3(B.lin_mom[0][v], B.lin_mom[1][v], B.lin_mom[2][v]);
    d3 am = mk3(B.ang_mom[0][v], B.ang_mom[1][v], B.ang_mom[2][v]);
    dq ang = mkq(B.quat[0][v], B.quat[1][v], B.quat[2][v], B.quat[3][v]);
    const double scale = B.scale[cur][v];
    d3 vel = lm * C.mass_inv;                 // Vel as left by the previous EulerStep (VXS_Voxel.cpp:409)

    // CalcTotalForce / CalcTotalMoment: fixed order PX,NX,PY,NY,PZ,NZ (VXS_Voxel.cpp:496-501,659-665)
    d3 F = (vel * (-R.slow_z)) * C.c_lin;
    d3 M = mk3(0, 0, 0);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (B.nbr[(2 * a) * B.nv + v] >= 0) {
            const int s = a * B.nv + v;
            F = F + mk3(B.bout[0][s], B.bout[1][s], B.bout[2][s]);
            M = M - mk3(B.bout[3][s], B.bout[4][s], B.bout[5][s]);
        }
        const int n = B.nbr[(2 * a + 1) * B.nv + v];
        if (n >= 0) {
            const int s = a * B.nv + n;
            F = F + mk3(B.bout[6][s], B.bout[7][s], B.bout[8][s]);
            M = M - mk3(B.bout[9][s], B.bout[10][s], B.bout[11][s]);
        }
    }
    if (flags & RF_SELF_COL) {                // collision bonds in creation order (VXS_Voxel.cpp:519-530)
        const int so = B.surf_ord[v];
        if (so >= 0) {
            const int row = R.surf_begin + so;
            const int cnt = B.col_cnt[row];
            for (int k = 0; k < cnt; ++k) {
                const int o = B.col_partner[row * VXH_MAXCOL + k];
                const int va = o < v ? o : v, vb = o < v ? v : o;      // Vox1 = earlier surface voxel
                d3 d = mk3(B.pos[cur][0][vb] - B.pos[cur][0][va], B.pos[cur][1][vb] - B.pos[cur][1][va], B.pos[cur][2][vb] - B.pos[cur][2][va]);
                const double nom = (B.scale[cur][va] + B.scale[cur][vb]) * 0.75;
                const double l = sqrt(len2(d));
                const double reld = nom - l;
                if (reld > 0) {
                    const DVoxClass& Co = B.vclass_tab[B.vclass[o]];
                    const double E = (C.E * Co.E / (C.E + Co.E)) * 2;
                    const double L = (C.nom_size + Co.nom_size) * 0.5;
                    const double a1 = E * (L * L) / L;
                    d3 f2 = ((d * (1.0 / l)) * a1) * reld;             // force on Vox2
                    F = (v == vb) ? F + f2 : F - f2;
                }
            }
        }
    }
    if ((flags & RF_GRAV) && !fluid) F.z += C.mass * R.grav_acc;

    if ((flags & RF_FLOOR) && !fluid) {       // CalcFloorEffect, VXS_Voxel.cpp:708-758
        const double pen = 0.5 * scale - pos.z;
        bool static_fric = false;
        if (pen > 0) {
            const double normal = C.k_floor * pen;
            double fz = normal - R.col_z * C.c_lin * vel.z;
            const double surf_vel = sqrt(vel.x * vel.x + vel.y * vel.y);
            const double surf_force = sqrt(F.x * F.x + F.y * F.y);
            const double fric = C.u_dynamic * normal;
            double fx = 0, fy = 0;
            bool stopped = (vel.x == 0 && vel.y == 0);
            if (flags & RF_STICKY) { lm.x = 0; lm.y = 0; static_fric = true; stopped = true; }
            if (stopped) {
                if (surf_force < C.u_static * normal) static_fric = true;
            } else if (fric * dt < C.mass * surf_vel) {
                // -(cos, sin)(atan2(vy, vx)) * fric == -(vx, vy)/|v| * fric
                const double inv = fric / surf_vel;
                fx = -vel.x * inv; fy = -vel.y * inv;
            } else { static_fric = true; lm.x = 0; lm.y = 0; }
            F.x += fx; F.y += fy; F.z += fz;
        }
        if (static_fric) { F.x = 0; F.y = 0; }
    }

    // EulerStep, VXS_Voxel.cpp:183-222
    lm = lm + F * dt;
    pos = pos + lm * (dt * C.mass_inv);
    am = am + M * dt;
    const double amf = 1 - 10 * R.slow_z * C.inertia_inv * C.c_ang * dt;
    am = am * amf;
    d3 w = am * C.inertia_inv;
    dq spin = qmul(mkq(0, w.x * 0.5, w.y * 0.5, w.z * 0.5), ang);
    ang = mkq(ang.w + spin.w * dt, ang.x + spin.x * dt, ang.y + spin.y * dt, ang.z + spin.z * dt);
    {
        const double l = sqrt(ang.x * ang.x + ang.y * ang.y + ang.z * ang.z + ang.w * ang.w);
        if (l != 0) { const double li = 1.0 / l; ang.w *= li; ang.x *= li; ang.y *= li; ang.z *= li; }
        if (ang.w >= 1.0) ang = mkq(1.0, 0, 0, 0);
    }

    // thermal actuation -> new scale (VXS_Voxel.cpp:224-340 without development; LW/VXS_Voxel.cpp:211-235)
    double new_scale;
    const double t = rs.cur_time;
    const double two_pi_f = (double)(2 * 3.1415926f);
    if (!(flags & RF_LW)) {
        const double c = (t >= 0.5 * R.init_cm_time) ? 1.0 : 2 * t / R.init_cm_time;
        const double prenatal = c * (((float)C.nom_size / C.nom_size) - 1);
        double ctrl = 0;
        if ((flags & RF_TEMP) && t >= R.init_cm_time)
            ctrl = (double)B.amp_damp[v] * ((double)R.temp_amplitude * sin(two_pi_f * (t / (double)R.temp_period + (double)B.phase[v]))) * C.cte;
        new_scale = ctrl * C.nom_size + (1 + prenatal) * C.nom_size;
        const double max_scale = (1 + R.growth_amplitude) * C.nom_size, min_scale = R.min_temp_fact * C.nom_size;
        if (new_scale < scale && new_scale < min_scale) new_scale = scale;
        if (new_scale > scale && new_scale > max_scale) new_scale = scale;
    } else {
        double tf = 1.0;
        if ((flags & RF_TEMP) && t >= R.init_cm_time)
            tf = 1 + ((double)R.temp_amplitude * sin(two_pi_f * (t / (double)R.temp_period + (double)B.phase[v]))) * C.cte;
        if (tf < 0.1) tf = 0.1;
        new_scale = tf * C.nom_size;
    }

    B.pos[nxt][0][v] = pos.x; B.pos[nxt][1][v] = pos.y; B.pos[nxt][2][v] = pos.z;
    B.scale[nxt][v] = new_scale;
    B.lin_mom[0][v] = lm.x; B.lin_mom[1][v] = lm.y; B.lin_mom[2][v] = lm.z;
    B.ang_mom[0][v] = am.x; B.ang_mom[1][v] = am.y; B.ang_mom[2][v] = am.z;
    B.quat[0][v] = ang.w; B.quat[1][v] = ang.x; B.quat[2][v] = ang.y; B.quat[3][v] = ang.z;

    vel2_new = len2(lm * C.mass_inv);
    }  // valid

    if (R.flags & RF_SELF_COL) {              // SS.MaxVoxVel for the collision horizon (VX_Sim.cpp:1625-1649)
        double v2 = vel2_new;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { double o = __shfl_xor(v2, off); v2 = o > v2 ? o : v2; }
        if ((threadIdx.x & 63) == 0) atomicMax(&rs.maxvel2_bits, (unsigned long long)__double_as_longlong(v2));
    }
}

// ------------------------------------------------------------------------------------- per-robot step control
// one workgroup per robot; runs before the bond kernel of every step and once more after the last step
__global__ __launch_bounds__(256) void k_step_begin(DBatch B, long long step_cap, int begin_new_step)
{
    const int r = blockIdx.x;
    const DRobot& R = B.robot[r];
    DRobotState& rs = B.rstate[r];
    __shared__ double sh[4][256];
    __shared__ int s_go, s_latch, s_eol, s_rebuild, s_overflow;
    const int tid = threadIdx.x;

    if (tid == 0) {
        s_go = 0; s_latch = 0; s_eol = 0; s_rebuild = 0; s_overflow = 0;
        if (rs.status == 0) {
            if (rs.active) {                 // finish the step the previous launches computed
                if (rs.diverged) rs.status = 2;
                else { rs.cur_time += R.dt; rs.steps += 1; rs.dt_prev = R.dt; }
                rs.active = 0;
            }
            if (rs.status == 0) {
                const double t = rs.cur_time;
                bool stop = false;           // StopConditionMet, VX_Sim.cpp:1398-1423 (LW/VX_Sim.cpp:1160-1172)
                if ((R.flags & RF_LW) || !(t <= R.init_cm_time)) {
                    if (R.stop_type == 1) stop = rs.steps > (int)(R.stop_value + 0.5);
                    else if (R.stop_type == 2) stop = t > (R.stop_value + R.afterlife);
                    else if (R.stop_type == 3) stop = t > R.temp_period_d * R.stop_value;
                }
                if (stop) rs.status = 1;
                else if (begin_new_step && (long long)rs.steps < step_cap) {
                    s_go = 1;
                    if (!rs.cm_init && t > R.init_cm_time) s_latch = 1;                     // VX_Sim.cpp:1064
                    if (!(R.flags & RF_LW) && t >= R.stop_value && rs.eol_post_y == 0) s_eol = 1;   // :1078
                    if (R.flags & RF_SELF_COL) {                                                // UpdateCollisions :1729-1755
                        const double mv = sqrt(__longlong_as_double((long long)rs.maxvel2_bits));
                        rs.max_disp += fabs(mv * rs.dt_prev / R.lat);
                        rs.maxvel2_bits = 0ull;
                        if (!(R.flags & RF_HORIZON_COL) || rs.max_disp > (R.col_horizon - 1.0) / 2) { s_rebuild = 1; rs.max_disp = 0.0; }
                    }
                }
            }
        }
    }
    __syncthreads();
    if (!s_go) return;
    const int cur = rs.steps & 1;
    const int base = R.vox_begin;

    if (s_latch || s_eol) {
        // IniCM = SS.CurCM of the previous step: mass-weighted SEQUENTIAL sum in voxel order (GetCM, VX_Sim.cpp:2415-2430)
        // staged through LDS in 256-voxel chunks so that thread 0 adds in exactly the reference order
        double sx = 0, sy = 0, sz = 0, sm = 0, miny = 100000.0;
        for (int c0 = 0; c0 < R.nvox; c0 += 256) {
            const int i = c0 + tid;
            if (i < R.nvox) {
                const DVoxClass& C = B.vclass_tab[B.vclass[base + i]];
                sh[0][tid] = B.pos[cur][0][base + i]; sh[1][tid] = B.pos[cur][1][base + i]; sh[2][tid] = B.pos[cur][2][base + i];
                sh[3][tid] = (C.mat == 5) ? -C.mass : C.mass;   // sign bit marks the material excluded from PosteriorY
            }
            __syncthreads();
            if (tid == 0) {
                const int n = min(256, R.nvox - c0);
                for (int k = 0; k < n; ++k) {
                    const double m = fabs(sh[3][k]);
                    sx = __dadd_rn(sx, __dmul_rn(sh[0][k], m)); sy = __dadd_rn(sy, __dmul_rn(sh[1][k], m)); sz = __dadd_rn(sz, __dmul_rn(sh[2][k], m)); sm += m;
                    if (!(sh[3][k] < 0)) { const double y = sh[1][k] / R.lat; if (y < miny) miny = y; }
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            if (s_latch) { const double inv = 1.0 / sm; rs.ini_cm[0] = inv * sx; rs.ini_cm[1] = inv * sy; rs.ini_cm[2] = inv * sz; rs.cm_init = 1; }
            if (s_eol) rs.eol_post_y = miny;   // getPosteriorY, VX_Sim.cpp:2640-2656
        }
    }

    if (s_rebuild) {
        // CalcL1Bonds (VX_Sim.cpp:2357-2413): all surface pairs i<j; each surface voxel builds its own partner list
        // in ascending partner order, which is the creation order of its collision bonds in the reference
        const double H = R.col_horizon;
        for (int i = tid; i < R.nsurf; i += 256) {
            const int vi = B.surf[R.surf_begin + i];
            const d3 pi = mk3(B.pos[cur][0][vi], B.pos[cur][1][vi], B.pos[cur][2][vi]);
            const double si = B.scale[cur][vi];
            const int nb = B.near_off[vi], ne = B.near_off[vi + 1];
            int cnt = 0;
            for (int j = 0; j < R.nsurf; ++j) {
                if (j == i) continue;
                const int vj = B.surf[R.surf_begin + j];
                const d3 d = pi - mk3(B.pos[cur][0][vj], B.pos[cur][1][vj], B.pos[cur][2][vj]);
                const double d2 = len2(d);
                if (!(d2 < R.filter_dist2)) continue;
                int lo = nb, hi = ne - 1; bool near = false;      // !pV1->IsNearbyVox(SIndex2)
                while (lo <= hi) { const int mid = (lo + hi) >> 1; const int x = B.near_idx[mid]; if (x == vj) { near = true; break; } if (x < vj) lo = mid + 1; else hi = mid - 1; }
                if (near) continue;
                const double s1 = (j > i) ? si : B.scale[cur][vj];   // scale of Vox1 = the earlier one, used twice (:2382)
                const double act = H * (s1 + s1) * 0.5;
                if (d2 < act * act) {
                    if (cnt < VXH_MAXCOL) B.col_partner[(R.surf_begin + i) * VXH_MAXCOL + cnt] = vj;
                    ++cnt;
                }
            }
            if (cnt > VXH_MAXCOL) { cnt = VXH_MAXCOL; s_overflow = 1; }
            B.col_cnt[R.surf_begin + i] = cnt;
        }
        __syncthreads();
        if (tid == 0) { rs.rebuilds += 1; if (s_overflow) rs.col_overflow = 1; }
    }
    if (tid == 0) rs.active = 1;
}

}  // namespace vxh
