// Wide path of the batched Voxelyze stepper: k_robot_wide<BLOCK, MESH, TABG>, one workgroup per robot, the robot resident in its CU
// for a whole launch like k_robot_steps (kernels_fused.hpp) -- but laid out for the LATENCY of a step instead of its instruction
// count: a small robot (up to BLOCK voxels, up to 2 * BLOCK bonds) gets more threads than it has voxels, so that
//   * every bond has a lane of its own and ALL THREE AXES are evaluated in one round (the resident kernel runs three rounds, one
//     per axis, a barrier between them: three dependent ~650-instruction FP64 chains per step on wavefronts that have a SIMD to
//     themselves and issue a dependent FP64 instruction only every ~7 cycles, scripts/ubench/fp64_issue.hip).  The combined bond list
//     (DBatch::wlist) carries the axis per entry (bond_compute_rt); thread t takes entries t and t + BLOCK;
//   * a bond's outputs go to a RECORD of its own in LDS (Force1, -Moment1, Force2, -Moment2: one writer, no accumulation, no
//     barrier between bonds), and a voxel adds up the records of its six directions in the reference's own order
//     +X -X +Y -Y +Z -Z after the slow-damping term (CalcTotalForce / CalcTotalMoment, VXS_Voxel.cpp:496-501, 659-665) -- the
//     resident kernel sums per bond end instead (DESIGN.md "Numerics");
//   * when the robot has at most BLOCK / 2 voxels the two independent halves of a voxel's update run on DIFFERENT wavefronts
//     (voxel_update_lin on thread v, voxel_update_ang on thread BLOCK / 2 + v: translation with contacts, floor and friction
//     here, rotation + actuation there), each keeping its own momentum in registers.
// Everything else -- control thread, IniCM latch, broad-phase, contact rows in LDS, fluid drag -- is the resident kernel's code.
// Measured (round 3, 64 robots = a quarter of the CUs busy): DESIGN.md "Wide kernel".
// Dynamic LDS (doubles):
//   ps   [8][BLOCK]   pose tile, as in the resident kernel
//   rec  [region]     bond records, VXH_WIDE_REC doubles each (13: twelve values + one of padding, 26 dwords -- a wavefront's
//                     64-bit accesses at that stride are conflict-free); record DRobot::wzidx stays zero: what the missing
//                     directions of a voxel point at.  Between steps the same memory is the scratch of latch / broad-phase /
//                     drag (12 * BLOCK doubles, then the velocities and the mesh vertices of a robot in a fluid), all below the zero record.
//   tabs              class tables (not TABG)
//   st   [6][BLOCK]   MESH: directional strains
//   cmask, rc_a1, rc_code   contact rows of colliding robots (rows_to_lds)
//   ps'  [8][BLOCK]   (two_tiles) second pose tile, at the end: a step's new poses go into the tile nobody reads
#pragma once

namespace vxh {

enum { VXH_WIDE_REC = 13, VXH_WIDE_STATIC_LDS = 480 };

// bond `entry` of the combined list: poses from the pose tile, history from / to HBM (L2), outputs into record `slot`
template <int BLOCK, bool MESH>
__device__ __forceinline__ bool wide_bond(const DBatch& B, const DRobot& R, const DBondClass* bct, const double* ps, double* rec, int entry, int slot,
                                          unsigned& modebits, int shift, bool damp_on, double* st)
{
    unsigned nv = B.nv;
    asm volatile("" : "+s"(nv));              // (plane addresses rebuilt per bond by the scalar unit, see fused_bond)
    const int l1 = entry & 511, l2 = (entry >> 9) & 511, axis = (entry >> 18) & 3;
    BondHist H;                               // one 48-byte record per bond slot (DBatch::hist_aos), as in the resident kernel
    double2* const hrec = (double2*)(B.hist_aos + ((size_t)((unsigned)axis * nv) + (unsigned)(R.vox_begin + l1)) * 6);
    { const double2 h0 = hrec[0], h1 = hrec[1], h2 = hrec[2]; H.p0 = h0.x; H.p1 = h0.y; H.p2 = h1.x; H.g0 = h1.y; H.g1 = h2.x; H.g2 = h2.y; }
    H.flags = (modebits >> shift) & 3u;
    H.store_hist = false;
    const d3 p1 = mk3(ps[l1], ps[BLOCK + l1], ps[2 * BLOCK + l1]);
    const double s1 = ps[3 * BLOCK + l1];
    const dq q1 = mkq(ps[4 * BLOCK + l1], ps[5 * BLOCK + l1], ps[6 * BLOCK + l1], ps[7 * BLOCK + l1]);
    const d3 p2 = mk3(ps[l2], ps[BLOCK + l2], ps[2 * BLOCK + l2]);
    const double s2 = ps[3 * BLOCK + l2];
    const dq q2 = mkq(ps[4 * BLOCK + l2], ps[5 * BLOCK + l2], ps[6 * BLOCK + l2], ps[7 * BLOCK + l2]);
    const BondOut o = bond_compute_rt(axis, B, bct[(unsigned)entry >> 20], H, p1, q1, s1, p2, q2, s2, damp_on);
    if (H.store_hist) { hrec[0] = make_double2(H.p0, H.p1); hrec[1] = make_double2(H.p2, H.g0); hrec[2] = make_double2(H.g1, H.g2); }
    modebits = (modebits & ~(3u << shift)) | (H.flags << shift);
    if constexpr (MESH) {                     // SetStrainDir (VXS_BondInternal.cpp:300-304): +A side of voxel 1, -A side of voxel 2
        st[axis * BLOCK + l1] = o.strain1;
        st[(3 + axis) * BLOCK + l2] = o.strain2;
    }
    double* e = rec + slot * VXH_WIDE_REC;
    e[0] = o.f1.x; e[1] = o.f1.y; e[2] = o.f1.z; e[3] = -o.m1.x; e[4] = -o.m1.y; e[5] = -o.m1.z;
    e[6] = o.f2.x; e[7] = o.f2.y; e[8] = o.f2.z; e[9] = -o.m2.x; e[10] = -o.m2.y; e[11] = -o.m2.z;
    return o.diverged;
}

// acc += the three values at offset `off` of the records of the voxel's six directions (+X -X +Y -Y +Z -Z: the negative end of the
// bond in a + direction takes its Force1 / Moment1, offsets 0 / 3; the positive end of the bond in a - direction Force2 / Moment2,
// offsets 6 / 9).  g0: +X | -X << 10 | +Y << 20, g1: -Y | +Z << 10 | -Z << 20.  A missing direction reads the zero record.
__device__ __forceinline__ d3 wide_gather(const double* rec, int g0, int g1, int off, d3 acc)
{
    const int idx[6] = {g0 & 1023, (g0 >> 10) & 1023, (g0 >> 20) & 1023, g1 & 1023, (g1 >> 10) & 1023, (g1 >> 20) & 1023};
    double v[6][3];
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        const double* e = rec + idx[d] * VXH_WIDE_REC + off + ((d & 1) ? 6 : 0);
        v[d][0] = e[0]; v[d][1] = e[1]; v[d][2] = e[2];
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) acc = acc + mk3(v[d][0], v[d][1], v[d][2]);
    return acc;
}

template <int BLOCK, bool MESH, bool TABG>
__global__ __launch_bounds__(BLOCK, BLOCK / 256) void k_robot_wide(DBatch B, const DRobot* __restrict__ robots,
                                                                   const int* __restrict__ robot_list, long long step_cap, int iters,
                                                                   int lds_doubles_all, int two_tiles)
{
    extern __shared__ __align__(16) double lds[];
    // two_tiles: a second pose tile at the end of the dynamic LDS (the host grants it where the robot's layout leaves 8 * BLOCK
    // doubles: every walker, the smaller swimmers).  The new poses then go into the OTHER tile during the voxel phase -- nobody reads
    // it -- and the step has two workgroup barriers instead of three (see the step loop).
    const int lds_doubles = lds_doubles_all - (two_tiles ? 8 * BLOCK : 0);
    double* const ps_a = lds;
    double* const ps_b = two_tiles ? lds + lds_doubles : lds;
    double* ps = ps_a;                        // the tile that holds the poses at the start of the step
    double* const rec = lds + 8 * BLOCK;
    __shared__ DRobotState rs;
    __shared__ FusedCtl s_ctl[2];
    // ping-pong control words of the step loop (slot = step parity): a bond diverged in this step; max |v|^2 of this step's voxel
    // phase; MaxDispSinceLastBondUpdate before this step's collision-horizon update
    __shared__ int s_divf[2];
    __shared__ unsigned long long s_mv[2];
    __shared__ double s_disp[2];
    __shared__ int s_seg[2 * (BLOCK / 64)];
    static_assert(sizeof(DRobotState) + 2 * sizeof(FusedCtl) + 2 * sizeof(int) + 4 * sizeof(double) + 2 * 16 * sizeof(int) + 16 <= VXH_WIDE_STATIC_LDS, "static LDS bound");

    const int tid = threadIdx.x;
    const int r = __builtin_amdgcn_readfirstlane(robot_list[blockIdx.x]);
    const DRobot& R = robots[r];
    const unsigned nv = B.nv;
    const int base = R.vox_begin, nvox = R.nvox;
    // roles of this thread: the translation of voxel `tid` (valid), the rotation + size of voxel `la` (angr); the same voxel unless the
    // robot is small enough to give the two halves to different wavefronts
    const bool split = 2 * nvox <= BLOCK;
    // ... to wavefronts on OTHER SIMDs where that is possible (wavefront w sits on SIMD w % 4): the rotation halves take the
    // wavefronts from the top down -- 7, 6, 5, ... for voxels 0-63, 64-127, ... -- so that a robot of up to 192 voxels has its three
    // translation wavefronts (the slowest: floor and friction) on SIMDs 0 1 2 and its rotation wavefronts on SIMDs 3 2 1
    const int la = split ? ((BLOCK / 64 - 1 - (tid >> 6)) << 6) + (tid & 63) : tid;
    const bool valid = tid < nvox, angr = la >= 0 && la < nvox && (!split || tid >= BLOCK / 2);
    const int v = base + tid, va = base + la;
    if (tid == 0) rs = B.rstate[r];
    double* const tabs = rec + R.wregion;
    const int nbd = TABG ? 0 : R.n_bclass * (int)(sizeof(DBondClass) / 8), nvd = TABG ? 0 : R.n_vclass * (int)(sizeof(DVoxClass) / 8);
    const DBondClass* bct;
    const DVoxClass* vct;
    if constexpr (TABG) {
        bct = B.bclass_tab + R.btab_begin;
        vct = B.vclass_tab + R.vtab_begin;
    } else {
        for (int k = tid; k < nbd; k += BLOCK) tabs[k] = ((const double*)(B.bclass_tab + R.btab_begin))[k];
        for (int k = tid; k < nvd; k += BLOCK) tabs[nbd + k] = ((const double*)(B.vclass_tab + R.vtab_begin))[k];
        bct = (const DBondClass*)tabs;
        vct = (const DVoxClass*)(tabs + nbd);
    }
    double* const st = tabs + nbd + nvd;                       // MESH: [6][BLOCK]
    const int nvs = (nvox + 63) & ~63;
    double* const mesh = rec + 12 * BLOCK + 3 * nvs;           // a robot in a fluid: the voxels' velocities [3][nvs] and the mesh vertices, inside the record region, behind the scratch
    unsigned long long* const cmask = (unsigned long long*)(st + (MESH ? 6 * BLOCK : 0));
    double* const rc_a1 = (double*)cmask + BLOCK;
    const int pool_cap = (R.flags & RF_SELF_COL) ? max(0, (int)((lds_doubles - (int)(rc_a1 - lds)) * 2 / 3) - 1) : 0;
    int* const rc_code = (int*)(rc_a1 + pool_cap);
    if constexpr (MESH) {
#pragma unroll
        for (int k = 0; k < 6; ++k) st[k * BLOCK + tid] = valid ? B.strain[(unsigned)k * nv + (unsigned)v] : 0.0;
    }
    for (int k = tid; k < VXH_WIDE_REC; k += BLOCK) rec[R.wzidx * VXH_WIDE_REC + k] = 0.0;     // the zero record
    __syncthreads();

    // ---- this thread's roles: momenta -> registers, pose -> LDS; its bonds
    const DVoxClass& Cl = vct[valid ? B.vclass[v] : 0];
    const DVoxClass& Ca = vct[angr ? B.vclass[va] : 0];
    const int nb = R.wnbond;
    int e0 = tid < nb ? B.wlist[R.wl_begin + tid] : -1, e1 = tid + BLOCK < nb ? B.wlist[R.wl_begin + BLOCK + tid] : -1;
    unsigned modebits = 0;                    // 2 bits per bond: SmallAngle, history layout (DBatch::hist)
    if (e0 != -1) modebits |= (unsigned)(B.small_angle[(unsigned)((e0 >> 18) & 3) * nv + (base + (e0 & 511))] & 3);
    if (e1 != -1) modebits |= (unsigned)(B.small_angle[(unsigned)((e1 >> 18) & 3) * nv + (base + (e1 & 511))] & 3) << 2;
    int gl0 = 0, gl1 = 0, ga0 = 0, ga1 = 0;   // records of my voxel's six bonds (wide_gather), per role
    float amp_damp = 1.f;
    double ph_sin = 0, ph_cos = 1;
    d3 lm = mk3(0, 0, 0), am = mk3(0, 0, 0);
    {
        const int b0 = rs.steps & 1;
        if (valid) {
            gl0 = B.wgather[v]; gl1 = B.wgather[nv + v];
            lm = mk3(LINMOM(0, v), LINMOM(1, v), LINMOM(2, v));
            ps[tid] = POS(b0, 0, v); ps[BLOCK + tid] = POS(b0, 1, v); ps[2 * BLOCK + tid] = POS(b0, 2, v);
        }
        if (angr) {
            ga0 = B.wgather[va]; ga1 = B.wgather[nv + va];
            amp_damp = B.amp_damp[va]; ph_sin = B.act_sb[va]; ph_cos = B.act_cb[va];
            am = mk3(ANGMOM(0, va), ANGMOM(1, va), ANGMOM(2, va));
            ps[3 * BLOCK + la] = SCALE(b0, va);
            ps[4 * BLOCK + la] = QUAT(0, va); ps[5 * BLOCK + la] = QUAT(1, va); ps[6 * BLOCK + la] = QUAT(2, va); ps[7 * BLOCK + la] = QUAT(3, va);
        }
    }
    DragCache<BLOCK> dcache;
    if constexpr (MESH && DragCache<BLOCK>::KEEP) { if ((R.flags & RF_FLUID) && R.nmv > 0 && R.nfacet > 0) { dcache.load(B, R, tid); if (valid) { dcache.my_first = B.facet_first[v]; dcache.my_count = (int)B.facet_count[v]; } } }
    // my contact row (see k_robot_steps): partner count | (start of the LDS copy + 1) << VXH_ROWD_BITS
    int rowd = 0;
    auto rows_to_lds = [&]() {
        rowd = 0;
        if (!(R.flags & RF_SELF_COL)) return;
        const int tid_r = opaque_tid<BLOCK>();
        int row = -1;
        if (valid) { const int so = B.surf_ord[v]; if (so >= 0) row = R.surf_begin + so; }
        const int ccnt = row >= 0 ? B.col_cnt[row] : 0;
        if (pool_cap > 0) cmask[tid_r] = 0;
        __syncthreads();
        const int ccnt_l = ccnt <= 64 ? ccnt : 0;     // (a longer row stays in memory: 64 mask bits per voxel)
        int incl = ccnt_l;
        const int lane = tid_r & 63;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        const int wave_total = __shfl(incl, 63);
        if (lane == 0) s_seg[2 * (tid_r >> 6) + 1] = wave_total;
        __syncthreads();
        int wave_base = 0;
        for (int w = 0; w < (tid_r >> 6); ++w) wave_base += s_seg[2 * w + 1];
        __syncthreads();                      // (s_seg is rewritten below)
        const bool fits = wave_base + wave_total <= pool_cap;
        if (lane == 0) { s_seg[2 * (tid_r >> 6)] = wave_base; s_seg[2 * (tid_r >> 6) + 1] = fits ? wave_total : -1; }
        const int off = wave_base + incl - ccnt_l;
        rowd = ccnt;
        if (fits && ccnt_l > 0) {
            rowd = ccnt | ((off + 1) << VXH_ROWD_BITS);
            for (int k0 = 0; k0 < ccnt; k0 += 4) {           // (four entries' loads in flight)
                int pj[4]; double aj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const size_t at = col_at(R, min(k0 + j, ccnt - 1), row); pj[j] = B.col_partner[at]; aj[j] = B.col_a1[at]; }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + j < ccnt) { rc_code[off + k0 + j] = (pj[j] - base) | (tid_r << 10) | ((k0 + j) << 20); rc_a1[off + k0 + j] = aj[j]; }
            }
        }
        __syncthreads();
    };

    const bool ctl_thread = tid == (split ? 3 * 64 : BLOCK - 64);      // (split: wavefront 3 has the fewest voxels of either kind; else the last one)
    if (ctl_thread) {
        fused_control_begin(R, rs, step_cap, iters > 0, s_ctl[0]); fused_control_horizon(R, rs, s_ctl[0]);
        s_divf[0] = s_divf[1] = 0; s_mv[0] = s_mv[1] = 0ull; s_disp[0] = s_disp[1] = rs.max_disp;
    }
    rows_to_lds();
    __syncthreads();                           // control of the first step + every voxel's pose visible
    // wavefronts that finish the bond phase a bond early -- without a bond at all, or, when the list is longer than the workgroup, without a
    // SECOND one (the list fills the threads in order) -- run the contact reach test of the whole workgroup there (fused_contact_reach_all,
    // kernels_fused.hpp: the bits are read behind barrier (B))
    constexpr int NWAVES = BLOCK / 64;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbw = ((nb <= BLOCK ? nb : nb - BLOCK) + 63) >> 6;
    const bool reach_early = (R.flags & RF_SELF_COL) && nbw < NWAVES;
    auto pairs_in_copy = [&]() { int n = 0; for (int w = 0; w < NWAVES; ++w) n += max(s_seg[2 * w + 1], 0); return __builtin_amdgcn_readfirstlane(n); };
    int npairs = reach_early ? pairs_in_copy() : 0;
    // The step loop.  Barriers of a step: (B) behind the bond phase, (X) behind the voxel phase -- and, with ONE pose tile, (A) behind
    // the pose stores that then follow (X).  What used to sit between two barriers on one lane, the collision-horizon update
    // (UpdateCollisions: MaxDisp += |MaxVoxVel dt / lattice|, rebuild when it passes the horizon), every thread now evaluates for
    // itself behind (X) from the control words of the step's parity -- the same operations on the same operands, one result -- and
    // the control thread alone stores the next step's words.  Who reads and who writes a slot are always a barrier apart:
    //   s_mv[p]    atomicMax in voxel phase p (behind (B)) | read behind (X) of step p | zeroed behind (X) of step p - 1
    //   s_disp[p]  read behind (X) of step p | written behind (X) of step p - 1
    //   s_divf[p]  set in bond phase p | read behind (B) of step p | zeroed in voxel phase p - 1
    bool reb_next = false;                     // this step starts with a broad-phase run (decided behind (X) of the step before)
    long long steps_done = 0;
    VXH_T_DECL
    for (int it = 0;; ++it) {
        const int par = it & 1;
        const FusedCtl& K = s_ctl[par];
        FusedCtl& Knext = s_ctl[par ^ 1];
        double* const psn = ps == ps_a ? ps_b : ps_a;      // where the new poses go (the same tile without the second one)
        const FetchLds<BLOCK> fetch{ps, base};
        const int kf = __builtin_amdgcn_readfirstlane(K.flags);
        const bool k_go = kf & 1, k_latch = kf & 2, k_eol = kf & 4, k_rebuild = (kf & 8) || reb_next, k_trace = kf & 16;
        if (!k_go && !k_trace) break;
        int vv = v, vva = va;                  // opaque per-step copies (see k_robot_steps)
        asm volatile("" : "+v"(vv));
        asm volatile("" : "+v"(vva));
        if (k_latch || k_eol || k_trace)
            fused_latch_cm<BLOCK>(R, rs, ps, rec, valid, Cl, k_latch, k_eol, k_trace, B.trace + (size_t)(R.trace_begin + K.trace_index) * 4);
        if (!k_go) break;
        if (__builtin_expect(k_rebuild, 0)) {
            fused_rebuild<BLOCK>(B, R, rs, ps, rec, 12 * BLOCK, (double*)cmask, max(0, lds_doubles - (int)((double*)cmask - lds)), vct);
            rows_to_lds();
            if (reach_early) npairs = pairs_in_copy();
        }
        d3 drag = mk3(0, 0, 0);
        const bool fluid = MESH && (R.flags & RF_FLUID) != 0;
        if constexpr (MESH) { if (fluid) drag = fused_drag<BLOCK, 12 * BLOCK, true>(B, R, ps, st, (unsigned)BLOCK, mesh, rec, valid, vv, lm, Cl.mass_inv, dcache, rec + 12 * BLOCK, nvs); }
        const bool damp_on = (kf & 32) != 0;
        VXH_T_MARK(1)

        // ---- bond phase: every bond of the robot, all axes, one round (two for the threads that hold a second entry)
        bool div = false;
        if (e0 != -1) div = wide_bond<BLOCK, MESH>(B, R, bct, ps, rec, e0, tid, modebits, 0, damp_on, st);
        if (e1 != -1) div = wide_bond<BLOCK, MESH>(B, R, bct, ps, rec, e1, tid + BLOCK, modebits, 2, damp_on, st) || div;
        if (reach_early && wave >= nbw) fused_contact_reach_all<BLOCK>(ps, 0, npairs, wave - nbw, NWAVES - nbw, cmask, rc_code);
        if (div) s_divf[par] = 1;
        VXH_T_MARK(2)
        __syncthreads();                       // (B)
        VXH_T_MARK(3)
        if (s_divf[par]) {                     // Integrate() returns before the voxel loop (VX_Sim.cpp:1777)
            __syncthreads();
            if (ctl_thread) {
                rs.diverged = 1; fused_control_begin(R, rs, step_cap, 0, Knext);
                s_divf[par ^ 1] = 0; s_mv[par ^ 1] = 0ull; s_disp[par ^ 1] = s_disp[par];
            }
            reb_next = false;
            __syncthreads();
            continue;
        }
        // ---- voxel phase, translation: damping + bond forces in the reference's order, contacts, floor, integration
        double vel2 = 0;
        d3 pos = mk3(0, 0, 0);
        if ((R.flags & RF_SELF_COL) && !reach_early) {
            const int nseg = s_seg[2 * (tid >> 6) + 1];
            if (nseg > 0) fused_contact_reach<BLOCK>(ps, s_seg[2 * (tid >> 6)], nseg, cmask, rc_code);
        }
        if (valid) {
            const d3 vel = lm * Cl.mass_inv;
            d3 F = wide_gather(rec, gl0, gl1, 0, (vel * (-R.slow_z)) * Cl.c_lin);
            pos = mk3(ps[tid], ps[BLOCK + tid], ps[2 * BLOCK + tid]);
            const double scale = ps[3 * BLOCK + tid];
            if (rowd != 0) F = fused_contact_forces<BLOCK>(B, R, ps, F, pos, scale, tid, vv, rowd, cmask, rc_code, rc_a1);
            vel2 = voxel_update_lin(B, R, Cl, vv, fetch, F, vel, pos, lm, scale, -1, 0, fluid, drag);
        }
        // ---- voxel phase, rotation and size
        dq ang = mkq(1, 0, 0, 0);
        double scale_a = 0;
        if (angr) {
            const d3 M = wide_gather(rec, ga0, ga1, 3, mk3(0, 0, 0));
            scale_a = ps[3 * BLOCK + la];
            ang = mkq(ps[4 * BLOCK + la], ps[5 * BLOCK + la], ps[6 * BLOCK + la], ps[7 * BLOCK + la]);
            voxel_update_ang(B, R, Ca, vva, K.time, K.act_sin, K.act_cos, K.prenatal_c, M, am, ang, scale_a, ph_sin, ph_cos, amp_damp);
        }
        if (ctl_thread) { fused_control_begin(R, rs, step_cap, it + 1 < iters, Knext); s_divf[par ^ 1] = 0; }   // next step's control, off the critical path
        if (R.flags & RF_SELF_COL) {             // SS.MaxVoxVel for the collision horizon (VX_Sim.cpp:1625-1649)
            vel2 = wave_max_nonneg(vel2);
            if ((tid & 63) == 0) atomicMax(&s_mv[par], (unsigned long long)__double_as_longlong(vel2));
        }
        auto store_poses = [&]() {
            if (valid) { psn[tid] = pos.x; psn[BLOCK + tid] = pos.y; psn[2 * BLOCK + tid] = pos.z; }
            if (angr) {
                psn[3 * BLOCK + la] = scale_a;
                psn[4 * BLOCK + la] = ang.w; psn[5 * BLOCK + la] = ang.x; psn[6 * BLOCK + la] = ang.y; psn[7 * BLOCK + la] = ang.z;
            }
        };
        if (two_tiles) store_poses();          // (into the tile nobody reads)
        VXH_T_MARK(4)
        __syncthreads();                       // (X) the voxel phase is complete: max |v|^2 in, the next step's control block written
        if (!two_tiles) store_poses();         // (every read of the old poses is done)
        VXH_T_MARK(5)
        // the collision-horizon update of the NEXT step (step_control_horizon, kernels.hpp), by every thread for itself
        {
            const int nf = __builtin_amdgcn_readfirstlane(Knext.flags);
            bool reb = false;
            double disp = s_disp[par];
            if ((nf & 1) && (R.flags & RF_SELF_COL)) {
                const double mv = vsqrt_nn(__longlong_as_double((long long)s_mv[par]));
                disp += fabs(vdiv(mv * rs.dt_prev, R.lat));
                if (!(R.flags & RF_HORIZON_COL) || disp > (R.col_horizon - 1.0) / 2) { reb = true; disp = 0.0; }
            }
            reb_next = reb;
            if (ctl_thread) {
                s_disp[par ^ 1] = disp; s_mv[par ^ 1] = 0ull;
                if (reb) { rs.rebuilds += 1; rs.col_tiled = 0; }
                rs.rebuild_now = reb ? 1 : 0;
                // the words the control block carries between launches: the displacement sum as of now, and the max |v|^2 of this
                // step where the next step's update has not consumed it (the launch ends here)
                rs.max_disp = disp;
                rs.maxvel2_bits = ((nf & 1) && (R.flags & RF_SELF_COL)) ? 0ull : s_mv[par];
            }
        }
        ps = psn;
        ++steps_done;
        if (!two_tiles) __syncthreads();       // (A) every voxel's published pose visible
        VXH_T_MARK(0)
    }
    (void)steps_done;
    VXH_T_FLUSH
    // ---- back to HBM
    {
        const int b1 = rs.steps & 1;
        if (valid) {
            POS(b1, 0, v) = ps[tid]; POS(b1, 1, v) = ps[BLOCK + tid]; POS(b1, 2, v) = ps[2 * BLOCK + tid];
            LINMOM(0, v) = lm.x; LINMOM(1, v) = lm.y; LINMOM(2, v) = lm.z;
        }
        if (angr) {
            SCALE(b1, va) = ps[3 * BLOCK + la];
            QUAT(0, va) = ps[4 * BLOCK + la]; QUAT(1, va) = ps[5 * BLOCK + la]; QUAT(2, va) = ps[6 * BLOCK + la]; QUAT(3, va) = ps[7 * BLOCK + la];
            ANGMOM(0, va) = am.x; ANGMOM(1, va) = am.y; ANGMOM(2, va) = am.z;
        }
    }
    if constexpr (MESH) {
        if (valid) {
#pragma unroll
            for (int k = 0; k < 6; ++k) B.strain[(unsigned)k * nv + (unsigned)v] = st[k * BLOCK + tid];
        }
    }
    if (e0 != -1) B.small_angle[(unsigned)((e0 >> 18) & 3) * nv + (base + (e0 & 511))] = (unsigned char)(modebits & 3u);
    if (e1 != -1) B.small_angle[(unsigned)((e1 >> 18) & 3) * nv + (base + (e1 & 511))] = (unsigned char)((modebits >> 2) & 3u);
    if (tid == 0) { B.rstate[r] = rs; if (B.rstate_mirror) B.rstate_mirror[r] = rs; }
}

}  // namespace vxh
