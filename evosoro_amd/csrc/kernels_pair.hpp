// Resident path for the LARGE robots, two items per lane: k_robot_pair<TABG, SEL> (included at the end of kernels.hpp, behind
// kernels_fused.hpp whose device functions it shares).  One workgroup of 512 threads = 8 wavefronts, two per SIMD, steps one robot of up
// to 1024 voxels for a whole launch; thread t owns voxels t and t + 512 and, per axis, up to two bonds of the robot's compacted
// bond list.  Why (DESIGN.md section 4 "Pair path"): k_robot_steps<1024> needs 16 wavefronts = four per SIMD = 128 vector registers per
// lane, and at 128 the bond / voxel arithmetic spills (204-340 B of scratch per lane, reloaded inside the hot phases); its single
// accumulator tile forces every axis round into two barrier-separated sub-steps, five heavy barriers per bond phase with a dependent
// FP64 chain between each.  Eight wavefronts have 256 registers each: nothing spills, the outputs of a bond can WAIT in registers for
// their turn at the accumulators, and so the bonds of X and Y are evaluated in ONE barrier-free stretch of four bonds per lane (the
// chunks dealt so that the four SIMDs carry the same number), Z in a second one; the barriers that order the additions stand between
// a dozen LDS atomics each, not between bond evaluations.
//
// Order of the force sums = that of k_robot_steps<1024> (the reference's CalcTotalForce order, VXS_Voxel.cpp:482-530): an accumulator
// entry receives +X and -X (both added to the zero the voxel phase left: they commute), then +Y, -Y, +Z, -Z, each behind a barrier.
//
// Dynamic LDS (doubles):  ps [8][1024] pose tile | class tables (not TABG) | acc [6][1024] | cmask [1024] | contact-row pool
// (acc .. end is one stretch: the scratch of the whole-robot passes -- CoM latch, broad-phase bit matrix -- between steps).
// Per lane in registers for the whole launch: both voxels' momenta, actuation phase sin / cos, six bond entries, mode bits.
#pragma once

namespace vxh {

// (VXH_PAIR_T / _NV / _NW / _NVW: device_types.hpp -- the host sizes the schedule with them whether or not the kernel is compiled in)

// IniCM latch + EndOfLifetimePosteriorY + trace point from the pose tile (fused_latch_cm with two voxels per lane)
__device__ __forceinline__ void pair_latch_cm(const DRobot& R, DRobotState& rs, const double* ps, double* sh, const DVoxClass* vct, int cls0, int cls1,
                                              bool latch, bool eol, bool trace, double* trace_entry)
{
    constexpr int T = VXH_PAIR_T, NV = VXH_PAIR_NV;
    const int tid = threadIdx.x;
    if (tid < R.nvox) { const DVoxClass& C = vct[cls0]; sh[tid] = (C.mat == 5) ? -C.mass : C.mass; }      // sign marks the material excluded from PosteriorY
    if (tid + T < R.nvox) { const DVoxClass& C = vct[cls1]; sh[tid + T] = (C.mat == 5) ? -C.mass : C.mass; }
    __syncthreads();
    if (tid == 0) {
#pragma clang fp contract(off)      // product and sum rounded separately, like the reference's GetCM
        double sx = 0, sy = 0, sz = 0, sm = 0, miny = 100000.0;
        for (int k = 0; k < R.nvox; ++k) {
            const double ms = sh[k], m = fabs(ms), y = ps[NV + k];
            const double mx = ps[k] * m, my = y * m, mz = ps[2 * NV + k] * m;
            sx = sx + mx; sy = sy + my; sz = sz + mz; sm += m;
            if (!(ms < 0)) { const double yl = y / R.lat; if (yl < miny) miny = yl; }
        }
        if (latch) { const double inv = 1.0 / sm; rs.ini_cm[0] = inv * sx; rs.ini_cm[1] = inv * sy; rs.ini_cm[2] = inv * sz; rs.cm_init = 1; }
        if (eol) rs.eol_post_y = miny;
        if (trace) { const double inv = 1.0 / sm; trace_entry[0] = rs.cur_time; trace_entry[1] = inv * sx; trace_entry[2] = inv * sy; trace_entry[3] = inv * sz; }
    }
    __syncthreads();
}

// CalcL1Bonds (VX_Sim.cpp:2357-2413) as the bit matrix of fused_rebuild_sym (kernels_fused.hpp: the method, its exactness and the
// layout of `mat` / `stg` are described there), for 512 threads and up to 1024 surface voxels: staging and row emission walk the
// ordinals in strides of the workgroup, the block pairs are handed out to the wavefronts as there.
__device__ __forceinline__ void pair_rebuild_sym(const DBatch& B, const DRobot& R, DRobotState& rs, const double* ps, unsigned long long* mat,
                                                 double* stg, const DVoxClass* vct)
{
    constexpr int T = VXH_PAIR_T, NV = VXH_PAIR_NV;
    const int tid = threadIdx.x, ns = R.nsurf;
    const int nb = (ns + 63) >> 6, NS = nb << 6;
    const int lane = tid & 63;
    double* const box = stg + 3 * NS;
    int* const next_pair = (int*)(box + 8 * nb);
    int* const shi = next_pair + 4;
    double* const a1tab = box + 8 * nb + 2 + (nb << 5);
    const int nvc = R.n_vclass;
    const bool tab = nvc <= VXH_A1TAB_CLASSES;
    if (tab && tid < nvc * nvc) a1tab[tid] = contact_a1(vct[tid / nvc], vct[tid % nvc]);
    const double H = R.col_horizon, filter2 = R.filter_dist2;
    if (tid == 0) *next_pair = 0;
    for (int k = tid; k < NS; k += T) {                          // (whole wavefronts: NS and T are multiples of 64; a wavefront stages block k >> 6)
        double x = 1.0e150, y = 1.0e150, z = 1.0e150, thr = -1.0;
        int code = 0;
        if (k < ns) {
            code = B.surf_code[R.surf_begin + k];
            const int l = code & 1023;
            x = ps[l]; y = ps[NV + l]; z = ps[2 * NV + l];
            const double sk = ps[3 * NV + l];
            const double act = H * (sk + sk) * 0.5, act2 = act * act;
            thr = act2 < filter2 ? act2 : filter2;
        }
        stg[k] = x; stg[NS + k] = y; stg[2 * NS + k] = z; shi[k] = code;
        const bool real = k < ns;
        const double lox = wave_minmax<false>(real ? x : 1.0e300), loy = wave_minmax<false>(real ? y : 1.0e300), loz = wave_minmax<false>(real ? z : 1.0e300);
        const double hix = wave_minmax<true>(real ? x : -1.0e300), hiy = wave_minmax<true>(real ? y : -1.0e300), hiz = wave_minmax<true>(real ? z : -1.0e300);
        const double tmax = wave_minmax<true>(thr);
        if (lane == 0) { double* e = box + 8 * (k >> 6); e[0] = lox; e[1] = loy; e[2] = loz; e[3] = hix; e[4] = hiy; e[5] = hiz; e[6] = tmax; }
    }
    __syncthreads();
    const int npairs = nb * (nb + 1) / 2;
    for (;;) {
        int p = 0;
        if (lane == 0) p = atomicAdd(next_pair, 1);
        p = __builtin_amdgcn_readfirstlane(p);
        if (p >= npairs) break;
        int I = 0, rem = p;
        while (rem >= nb - I) { rem -= nb - I; ++I; }
        const int J = I + rem;
        const int i = (I << 6) + lane, jrow = (J << 6) + lane;
        unsigned long long e_own = ~0ull, e_col = ~0ull;
        if (i < ns) e_own = B.excl[R.excl_begin + (long long)i * R.excl_wpr + J];
        if (jrow < ns) e_col = B.excl[R.excl_begin + (long long)jrow * R.excl_wpr + I];
        if (I != J) {
            const double* bi = box + 8 * I; const double* bj = box + 8 * J;
            double gap2 = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double g1 = bi[a] - bj[3 + a], g2 = bj[a] - bi[3 + a];
                const double g = g1 > g2 ? g1 : g2;
                if (g > 0) gap2 += g * g;
            }
            if (gap2 > bi[6] * (1.0 + 1.0e-12)) {
                mat[(size_t)J * NS + i] = 0ull;
                mat[(size_t)I * NS + jrow] = 0ull;
                continue;
            }
        }
        const d3 pi = mk3(stg[i], stg[NS + i], stg[2 * NS + i]);
        double thr = -1.0;
        if (i < ns) {
            const double si = ps[3 * NV + (shi[i] & 1023)];
            const double act = H * (si + si) * 0.5;
            const double act2 = act * act;
            thr = act2 < filter2 ? act2 : filter2;
        }
        unsigned own_w[2] = {0, 0}, col_lo = 0, col_hi = 0;
        const double* q = stg + (J << 6);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            unsigned own = 0;
#pragma unroll 1
            for (int u0 = 0; u0 < 32; u0 += 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int u = 32 * half + u0 + k;
                    const d3 d = pi - mk3(q[u], q[NS + u], q[2 * NS + u]);
                    const bool in = len2(d) < thr;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
                    own |= in ? (1u << (u0 + k)) : 0u;
                    col_lo = writelane_u32(col_lo, (unsigned)m, u);
                    col_hi = writelane_u32(col_hi, (unsigned)(m >> 32), u);
                }
            }
            own_w[half] = own;
        }
        const unsigned long long own = ((unsigned long long)own_w[1] << 32) | own_w[0], col = ((unsigned long long)col_hi << 32) | col_lo;
        if (I == J) {
            const unsigned long long below = (1ull << lane) - 1ull;
            mat[(size_t)I * NS + i] = ((own & ~(below | (1ull << lane))) | (col & below)) & ~e_own;
        } else {
            mat[(size_t)J * NS + i] = own & ~e_own;
            mat[(size_t)I * NS + jrow] = col & ~e_col;
        }
    }
    __syncthreads();
    for (int i = tid; i < ns; i += T) {
        const int ci = shi[i] >> 10;
        const DVoxClass& Ci = vct[ci];
        int cnt = 0;
        for (int w = 0; w < nb; ++w) {
            unsigned long long word = mat[(size_t)w * NS + i];
            while (word) {
                const int u = __builtin_ctzll(word);
                word &= word - 1;
                const int j = (w << 6) + u;
                if (cnt < R.col_cap) {
                    const int other = shi[j], cj = other >> 10;
                    const size_t at = col_at(R, cnt, R.surf_begin + i);
                    B.col_partner[at] = R.vox_begin + (other & 1023);
                    double a1;
                    if (tab) a1 = (j > i) ? a1tab[ci * nvc + cj] : a1tab[cj * nvc + ci];
                    else { const DVoxClass& Cj = vct[cj]; a1 = (j > i) ? contact_a1(Ci, Cj) : contact_a1(Cj, Ci); }
                    B.col_a1[at] = a1;
                }
                ++cnt;
            }
        }
        if (cnt > R.col_cap) { cnt = R.col_cap; atomicOr(&rs.col_overflow, 1); }
        B.col_cnt[R.surf_begin + i] = cnt;
    }
    __syncthreads();
}

// ... and the staged row scan (fused_rebuild_staged) for a robot whose matrix does not fit the scratch: thread i walks all candidates of
// its rows i, i + 512.  `shi`: 1024 ints, then (one plane of doubles on) four planes of 1024 doubles.
__device__ __forceinline__ void pair_rebuild_staged(const DBatch& B, const DRobot& R, DRobotState& rs, const double* ps, int* shi, const DVoxClass* vct)
{
    constexpr int T = VXH_PAIR_T, NV = VXH_PAIR_NV;
    const int tid = threadIdx.x, ns = R.nsurf;
    double* const stg = (double*)shi + NV;
    for (int k = tid; k < NV; k += T) {
        if (k < ns) {
            const int code = B.surf_code[R.surf_begin + k], l = code & 1023;
            shi[k] = code;
            stg[k] = ps[l]; stg[NV + k] = ps[NV + l]; stg[2 * NV + k] = ps[2 * NV + l]; stg[3 * NV + k] = ps[3 * NV + l];
        } else { stg[k] = 1.0e150; stg[NV + k] = 1.0e150; stg[2 * NV + k] = 1.0e150; stg[3 * NV + k] = 0.0; }     // (a group of eight may reach past ns)
    }
    __syncthreads();
    for (int i = tid; i < ns; i += T) {
        const int mine = shi[i];
        const DVoxClass& Ci = vct[mine >> 10];
        const d3 pi = mk3(stg[i], stg[NV + i], stg[2 * NV + i]);
        const double si = stg[3 * NV + i];
        const unsigned long long* row = B.excl + R.excl_begin + (long long)i * R.excl_wpr;
        const double H = R.col_horizon, filter2 = R.filter_dist2;
        int cnt = 0;
        unsigned long long word = 0;
        for (int j0 = 0; j0 < ns; j0 += 8) {
            if ((j0 & 63) == 0) word = row[j0 >> 6];
            int any = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = min(j0 + u, NV - 1);
                const d3 d = pi - mk3(stg[j], stg[NV + j], stg[2 * NV + j]);
                any |= (int)(len2(d) < filter2) & (int)(j != i) & (int)(j < ns);
            }
            if (!any) continue;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                if (j >= ns || j == i) continue;
                const d3 d = pi - mk3(stg[j], stg[NV + j], stg[2 * NV + j]);
                const double d2 = len2(d);
                if (!(d2 < filter2)) continue;
                if ((word >> (j & 63)) & 1ull) continue;
                const double s1 = (j > i) ? si : stg[3 * NV + j];
                const double act = H * (s1 + s1) * 0.5;
                if (d2 < act * act) {
                    if (cnt < R.col_cap) {
                        const int other = shi[j];
                        const DVoxClass& Cj = vct[other >> 10];
                        const size_t at = col_at(R, cnt, R.surf_begin + i);
                        B.col_partner[at] = R.vox_begin + (other & 1023);
                        B.col_a1[at] = (j > i) ? contact_a1(Ci, Cj) : contact_a1(Cj, Ci);
                    }
                    ++cnt;
                }
            }
        }
        if (cnt > R.col_cap) { cnt = R.col_cap; atomicOr(&rs.col_overflow, 1); }
        B.col_cnt[R.surf_begin + i] = cnt;
    }
    __syncthreads();
}

// A bond of axis A (packed entry: negative-end voxel | positive-end voxel << 10 | class << 20): fused_bond with the pose tile's stride
// and the rotation-vector form as template arguments of this kernel
template <int A, bool SEL>
__device__ __forceinline__ BondOut pair_bond(const DBatch& B, const DRobot& R, const DBondClass* bct, const double* ps, int entry, unsigned& modebits, bool damp_on)
{
    constexpr int NV = VXH_PAIR_NV;
    unsigned nv = B.nv;
    asm volatile("" : "+s"(nv));
    const int l1 = entry & 1023, l2 = (entry >> 10) & 1023;
    BondHist H;
    double2* const hrec = (double2*)(B.hist_aos + ((size_t)((unsigned)A * nv) + (unsigned)(R.vox_begin + l1)) * 6);
    { const double2 h0 = hrec[0], h1 = hrec[1], h2 = hrec[2]; H.p0 = h0.x; H.p1 = h0.y; H.p2 = h1.x; H.g0 = h1.y; H.g1 = h2.x; H.g2 = h2.y; }
    H.flags = (modebits >> (2 * A)) & 3u;
    H.store_hist = false;
    const d3 p1 = mk3(ps[l1], ps[NV + l1], ps[2 * NV + l1]);
    const double s1 = ps[3 * NV + l1];
    const dq q1 = mkq(ps[4 * NV + l1], ps[5 * NV + l1], ps[6 * NV + l1], ps[7 * NV + l1]);
    const d3 p2 = mk3(ps[l2], ps[NV + l2], ps[2 * NV + l2]);
    const double s2 = ps[3 * NV + l2];
    const dq q2 = mkq(ps[4 * NV + l2], ps[5 * NV + l2], ps[6 * NV + l2], ps[7 * NV + l2]);
    BondOut o = bond_compute<A, SEL>(B, bct[(unsigned)entry >> 20], H, p1, q1, s1, p2, q2, s2, damp_on);
    if (H.store_hist) { hrec[0] = make_double2(H.p0, H.p1); hrec[1] = make_double2(H.p2, H.g0); hrec[2] = make_double2(H.g1, H.g2); }
    modebits = (modebits & ~(3u << (2 * A))) | (H.flags << (2 * A));
    return o;
}

__device__ __forceinline__ void pair_add(double* acc, int l, d3 f, d3 m)
{
    constexpr int NV = VXH_PAIR_NV;
    double* e = acc + l;
    lds_add(e, f.x); lds_add(e + NV, f.y); lds_add(e + 2 * NV, f.z); lds_add(e + 3 * NV, -m.x); lds_add(e + 4 * NV, -m.y); lds_add(e + 5 * NV, -m.z);
}

template <bool TABG, bool SEL>
__global__ __launch_bounds__(VXH_PAIR_T, 2) void k_robot_pair(DBatch B, const DRobot* __restrict__ robots, const int* __restrict__ robot_list,
                                                              long long step_cap, int iters, int lds_doubles)
{
    constexpr int T = VXH_PAIR_T, NV = VXH_PAIR_NV, NW = VXH_PAIR_NW, NVW = VXH_PAIR_NVW;
    extern __shared__ __align__(16) double lds[];
    double* const ps = lds;
    double* const tabs = lds + 8 * NV;
    __shared__ DRobotState rs;
    __shared__ FusedCtl s_ctl[2];
    __shared__ int s_div, s_seg[2 * NVW];
    static_assert(sizeof(DRobotState) + 2 * sizeof(FusedCtl) + sizeof(int) + 2 * NVW * sizeof(int) + 16 <= VXH_FUSED_STATIC_LDS, "static LDS bound");

    const int tid = threadIdx.x;
    const int r = __builtin_amdgcn_readfirstlane(robot_list[blockIdx.x]);
    const DRobot& R = robots[r];
    const unsigned nv = B.nv;
    const int base = R.vox_begin;
    const bool valid0 = tid < R.nvox, valid1 = tid + T < R.nvox;
    if (tid == 0) rs = B.rstate[r];
    const int nbd = TABG ? 0 : R.n_bclass * (int)(sizeof(DBondClass) / 8), nvd = TABG ? 0 : R.n_vclass * (int)(sizeof(DVoxClass) / 8);
    const DBondClass* bct;
    const DVoxClass* vct;
    if constexpr (TABG) {
        bct = B.bclass_tab + R.btab_begin;
        vct = B.vclass_tab + R.vtab_begin;
    } else {
        for (int k = tid; k < nbd; k += T) tabs[k] = ((const double*)(B.bclass_tab + R.btab_begin))[k];
        for (int k = tid; k < nvd; k += T) tabs[nbd + k] = ((const double*)(B.vclass_tab + R.vtab_begin))[k];
        bct = (const DBondClass*)tabs;
        vct = (const DVoxClass*)(tabs + nbd);
    }
    double* const acc = tabs + ((nbd + nvd + 1) & ~1);
    unsigned long long* const cmask = (unsigned long long*)(acc + 6 * NV);
    double* const rc_a1 = (double*)cmask + NV;
    const int pool_cap = (R.flags & RF_SELF_COL) ? min((int)VXH_RIMG_CAP, max(0, (int)((lds_doubles - (int)(rc_a1 - lds)) * 2 / 3) - 1)) : 0;
    int* const rc_code = (int*)(rc_a1 + pool_cap);
    const int scratch_doubles = lds_doubles - (int)(acc - lds);          // acc .. end: the scratch of the whole-robot passes

    // ---- this thread's two voxels and six bonds
    const int cls0 = valid0 ? B.vclass[base + tid] : 0, cls1 = valid1 ? B.vclass[base + tid + T] : 0;
    int eX0, eX1, eY0, eY1, eZ0, eZ1;
    unsigned mb0 = 0, mb1 = 0;
    {
        const int* sc = B.bsched + R.sched_begin + tid;
        eX0 = sc[0]; eX1 = sc[T]; eY0 = sc[2 * T]; eY1 = sc[3 * T]; eZ0 = sc[4 * T]; eZ1 = sc[5 * T];
        if (eX0 != -1) mb0 |= (unsigned)(B.small_angle[base + (eX0 & 1023)] & 3);
        if (eX1 != -1) mb1 |= (unsigned)(B.small_angle[base + (eX1 & 1023)] & 3);
        if (eY0 != -1) mb0 |= (unsigned)(B.small_angle[nv + (base + (eY0 & 1023))] & 3) << 2;
        if (eY1 != -1) mb1 |= (unsigned)(B.small_angle[nv + (base + (eY1 & 1023))] & 3) << 2;
        if (eZ0 != -1) mb0 |= (unsigned)(B.small_angle[2u * nv + (base + (eZ0 & 1023))] & 3) << 4;
        if (eZ1 != -1) mb1 |= (unsigned)(B.small_angle[2u * nv + (base + (eZ1 & 1023))] & 3) << 4;
    }
    float amp0 = 1.f, amp1 = 1.f;
    double phs0 = 0, phc0 = 1, phs1 = 0, phc1 = 1;
    d3 lm0 = mk3(0, 0, 0), am0 = mk3(0, 0, 0), lm1 = mk3(0, 0, 0), am1 = mk3(0, 0, 0);
    __syncthreads();                           // rs and the tables are in
    {
        const int b0 = rs.steps & 1;
        if (valid0) {
            const int v = base + tid;
            amp0 = B.amp_damp[v]; phs0 = B.act_sb[v]; phc0 = B.act_cb[v];
            lm0 = mk3(LINMOM(0, v), LINMOM(1, v), LINMOM(2, v)); am0 = mk3(ANGMOM(0, v), ANGMOM(1, v), ANGMOM(2, v));
            ps[tid] = POS(b0, 0, v); ps[NV + tid] = POS(b0, 1, v); ps[2 * NV + tid] = POS(b0, 2, v); ps[3 * NV + tid] = SCALE(b0, v);
            ps[4 * NV + tid] = QUAT(0, v); ps[5 * NV + tid] = QUAT(1, v); ps[6 * NV + tid] = QUAT(2, v); ps[7 * NV + tid] = QUAT(3, v);
        }
        if (valid1) {
            const int v = base + tid + T, l = tid + T;
            amp1 = B.amp_damp[v]; phs1 = B.act_sb[v]; phc1 = B.act_cb[v];
            lm1 = mk3(LINMOM(0, v), LINMOM(1, v), LINMOM(2, v)); am1 = mk3(ANGMOM(0, v), ANGMOM(1, v), ANGMOM(2, v));
            ps[l] = POS(b0, 0, v); ps[NV + l] = POS(b0, 1, v); ps[2 * NV + l] = POS(b0, 2, v); ps[3 * NV + l] = SCALE(b0, v);
            ps[4 * NV + l] = QUAT(0, v); ps[5 * NV + l] = QUAT(1, v); ps[6 * NV + l] = QUAT(2, v); ps[7 * NV + l] = QUAT(3, v);
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) { acc[k * NV + tid] = 0.0; acc[k * NV + T + tid] = 0.0; }
    const FetchLds<NV> fetch{ps, base};

    // my two contact rows: partner count | (start of the LDS copy of the row + 1) << VXH_ROWD_BITS; s_seg: the segments of the sixteen
    // groups of 64 voxels (group g = voxels 64 g ..: lanes of wavefront g mod 8, voxel half g / 8), like the wavefronts' of k_robot_steps
    int rowd0 = 0, rowd1 = 0;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto rows_to_lds = [&](bool at_launch) {
        rowd0 = rowd1 = 0;
        if (!(R.flags & RF_SELF_COL)) return;
        const int img = R.img_index;
        if (at_launch && img >= 0 && rs.rows_img != 0) {
            const int* const seg = B.rimg_seg + (size_t)img * 64;
            const int used = __builtin_amdgcn_readfirstlane(seg[2 * NVW]);
            if (valid0) rowd0 = B.rimg_rowd[base + tid];
            if (valid1) rowd1 = B.rimg_rowd[base + tid + T];
            if (tid < 2 * NVW) s_seg[tid] = seg[tid];
            for (int k = tid; k < used; k += T) { rc_code[k] = B.rimg_code[(size_t)img * VXH_RIMG_CAP + k]; rc_a1[k] = B.rimg_a1[(size_t)img * VXH_RIMG_CAP + k]; }
            if (pool_cap > 0) { cmask[tid] = 0; cmask[tid + T] = 0; }
            __syncthreads();
            return;
        }
        int row0 = -1, row1 = -1;
        if (valid0) { const int so = B.surf_ord[base + tid]; if (so >= 0) row0 = R.surf_begin + so; }
        if (valid1) { const int so = B.surf_ord[base + tid + T]; if (so >= 0) row1 = R.surf_begin + so; }
        const int ccnt0 = row0 >= 0 ? B.col_cnt[row0] : 0, ccnt1 = row1 >= 0 ? B.col_cnt[row1] : 0;
        if (pool_cap > 0) { cmask[tid] = 0; cmask[tid + T] = 0; }
        __syncthreads();
        const int cl0 = ccnt0 <= 64 ? ccnt0 : 0, cl1 = ccnt1 <= 64 ? ccnt1 : 0;     // (a longer row stays in memory: 64 mask bits per voxel)
        int incl0 = cl0, incl1 = cl1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t0 = __shfl_up(incl0, d), t1 = __shfl_up(incl1, d); if (lane >= d) { incl0 += t0; incl1 += t1; } }
        const int tot0 = __shfl(incl0, 63), tot1 = __shfl(incl1, 63);
        if (lane == 0) { s_seg[2 * wave + 1] = tot0; s_seg[2 * (NW + wave) + 1] = tot1; }
        __syncthreads();
        int wb0 = 0, wb1 = 0, all_total = 0;
        for (int w = 0; w < NVW; ++w) { const int t = s_seg[2 * w + 1]; if (w < wave) wb0 += t; if (w < NW + wave) wb1 += t; all_total += t; }
        __syncthreads();                      // (s_seg is rewritten below)
        const bool fits0 = wb0 + tot0 <= pool_cap, fits1 = wb1 + tot1 <= pool_cap;
        if (lane == 0) { s_seg[2 * wave] = wb0; s_seg[2 * wave + 1] = fits0 ? tot0 : -1; s_seg[2 * (NW + wave)] = wb1; s_seg[2 * (NW + wave) + 1] = fits1 ? tot1 : -1; }
        rowd0 = ccnt0; rowd1 = ccnt1;
        auto copy_row = [&](int row, int ccnt, int off, int owner) {
            for (int k0 = 0; k0 < ccnt; k0 += 4) {
                int pj[4]; double aj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const size_t at = col_at(R, min(k0 + j, ccnt - 1), row); pj[j] = B.col_partner[at]; aj[j] = B.col_a1[at]; }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + j < ccnt) { rc_code[off + k0 + j] = (pj[j] - base) | (owner << 10) | ((k0 + j) << 20); rc_a1[off + k0 + j] = aj[j]; }
            }
        };
        if (fits0 && cl0 > 0) { const int off = wb0 + incl0 - cl0; rowd0 = ccnt0 | ((off + 1) << VXH_ROWD_BITS); copy_row(row0, ccnt0, off, tid); }
        if (fits1 && cl1 > 0) { const int off = wb1 + incl1 - cl1; rowd1 = ccnt1 | ((off + 1) << VXH_ROWD_BITS); copy_row(row1, ccnt1, off, tid + T); }
        __syncthreads();
        if (img >= 0) {                       // save the copy for the next launches (stores only: nothing waits for them)
            const int used = min(min(all_total, pool_cap), (int)VXH_RIMG_CAP);
            int* const seg = B.rimg_seg + (size_t)img * 64;
            if (valid0) B.rimg_rowd[base + tid] = rowd0;
            if (valid1) B.rimg_rowd[base + tid + T] = rowd1;
            if (tid < 2 * NVW) seg[tid] = s_seg[tid];
            if (tid == 0) { seg[2 * NVW] = used; rs.rows_img = (all_total <= VXH_RIMG_CAP || pool_cap <= VXH_RIMG_CAP) ? 1 : 0; }
            for (int k = tid; k < used; k += T) { B.rimg_code[(size_t)img * VXH_RIMG_CAP + k] = rc_code[k]; B.rimg_a1[(size_t)img * VXH_RIMG_CAP + k] = rc_a1[k]; }
        }
    };

    const bool ctl_thread = tid == T - 64;
    if (ctl_thread) { fused_control_begin(R, rs, step_cap, iters > 0, s_ctl[0]); fused_control_horizon(R, rs, s_ctl[0]); s_div = 0; }
    rows_to_lds(true);
    __syncthreads();                           // control of the first step + every voxel's pose visible
    const bool selfcol = (R.flags & RF_SELF_COL) != 0;
    auto pairs_in_copy = [&]() { int n = 0; for (int w = 0; w < NVW; ++w) n += max(s_seg[2 * w + 1], 0); return __builtin_amdgcn_readfirstlane(n); };
    int npairs = selfcol ? pairs_in_copy() : 0;
    VXH_T_DECL
    for (int it = 0;; ++it) {
        const FusedCtl& K = s_ctl[it & 1];
        FusedCtl& Knext = s_ctl[(it + 1) & 1];
        const int kf = __builtin_amdgcn_readfirstlane(K.flags);
        const bool k_go = kf & 1, k_latch = kf & 2, k_eol = kf & 4, k_rebuild = kf & 8, k_trace = kf & 16;
        if (!k_go && !k_trace) break;
        int tt = tid;
        asm volatile("" : "+v"(tt));          // opaque per-step copy: the step's addresses are not hoisted out of the loop (and then spilled)
        bool scratch_used = false;
        if (k_latch || k_eol || k_trace) {
            pair_latch_cm(R, rs, ps, acc, vct, cls0, cls1, k_latch, k_eol, k_trace, B.trace + (size_t)(R.trace_begin + K.trace_index) * 4);
            scratch_used = true;
        }
        if (!k_go) break;
        if (__builtin_expect(k_rebuild, 0)) {
            const int nb = (R.nsurf + 63) >> 6, NS = nb << 6;
            const int need_mat = nb * NS, need_stg = fused_sym_stage_doubles(nb);
            if (need_mat + need_stg <= scratch_doubles) pair_rebuild_sym(B, R, rs, ps, (unsigned long long*)acc, acc + need_mat, vct);
            else pair_rebuild_staged(B, R, rs, ps, (int*)acc, vct);
            rows_to_lds(false);
            npairs = pairs_in_copy();
            scratch_used = true;
        }
        if (scratch_used) {
#pragma unroll
            for (int k = 0; k < 6; ++k) { acc[k * NV + tt] = 0.0; acc[k * NV + T + tt] = 0.0; }
            __syncthreads();
        }
        const bool damp_on = (kf & 32) != 0;
        VXH_T_MARK(1)

        // ---- bond phase.  Stretch 1: my two X bonds (both ends straight into the accumulators: +X and -X commute on the zero the voxel
        // phase left) and my two Y bonds (outputs held).  Barrier; +Y ends in.  Stretch 2: my two Z bonds (held) and my share of the contact
        // reach test.  Then -Y, +Z, -Z, a barrier and a dozen LDS atomics each.
        bool div = false;
        BondOut y0, y1, z0, z1;
        // (the two bonds of an axis in a LOOP, not unrolled: six inlined copies of the bond arithmetic and two of the voxel update made a
        // step loop of ~60 KB of code for a 64 KB instruction cache that two CUs share -- first version of this kernel, 44 us per step on
        // dense 10^3 lattices against 34 of k_robot_steps<1024>)
#pragma nounroll
        for (int h = 0; h < 2; ++h) {
            const int e = h ? eX1 : eX0;
            if (e != -1) {
                unsigned mb = h ? mb1 : mb0;
                const BondOut o = pair_bond<0, SEL>(B, R, bct, ps, e, mb, damp_on);
                div = div || o.diverged;
                pair_add(acc, e & 1023, o.f1, o.m1); pair_add(acc, (e >> 10) & 1023, o.f2, o.m2);
                if (h) mb1 = mb; else mb0 = mb;
            }
        }
#pragma nounroll
        for (int h = 0; h < 2; ++h) {
            const int e = h ? eY1 : eY0;
            if (e != -1) {
                unsigned mb = h ? mb1 : mb0;
                const BondOut o = pair_bond<1, SEL>(B, R, bct, ps, e, mb, damp_on);
                div = div || o.diverged;
                if (h) { mb1 = mb; y1 = o; } else { mb0 = mb; y0 = o; }
            }
        }
        __syncthreads();                       // every X contribution is in
        if (eY0 != -1) pair_add(acc, eY0 & 1023, y0.f1, y0.m1);
        if (eY1 != -1) pair_add(acc, eY1 & 1023, y1.f1, y1.m1);
#pragma nounroll
        for (int h = 0; h < 2; ++h) {
            const int e = h ? eZ1 : eZ0;
            if (e != -1) {
                unsigned mb = h ? mb1 : mb0;
                const BondOut o = pair_bond<2, SEL>(B, R, bct, ps, e, mb, damp_on);
                div = div || o.diverged;
                if (h) { mb1 = mb; z1 = o; } else { mb0 = mb; z0 = o; }
            }
        }
        if (selfcol) fused_contact_reach_all<NV>(ps, 0, npairs, wave, NW, cmask, rc_code);
        __syncthreads();                       // +Y in
        if (eY0 != -1) pair_add(acc, (eY0 >> 10) & 1023, y0.f2, y0.m2);
        if (eY1 != -1) pair_add(acc, (eY1 >> 10) & 1023, y1.f2, y1.m2);
        __syncthreads();                       // -Y in
        if (eZ0 != -1) pair_add(acc, eZ0 & 1023, z0.f1, z0.m1);
        if (eZ1 != -1) pair_add(acc, eZ1 & 1023, z1.f1, z1.m1);
        __syncthreads();                       // +Z in
        if (eZ0 != -1) pair_add(acc, (eZ0 >> 10) & 1023, z0.f2, z0.m2);
        if (eZ1 != -1) pair_add(acc, (eZ1 >> 10) & 1023, z1.f2, z1.m2);
        if (div) s_div = 1;
        VXH_T_MARK(2)
        __syncthreads();                       // (B)
        VXH_T_MARK(3)
        if (s_div) {                           // Integrate() returns before the voxel loop (VX_Sim.cpp:1777)
            __syncthreads();
            if (ctl_thread) { rs.diverged = 1; fused_control_begin(R, rs, step_cap, 0, Knext); s_div = 0; }
#pragma unroll
            for (int k = 0; k < 6; ++k) { acc[k * NV + tt] = 0.0; acc[k * NV + T + tt] = 0.0; }
            if (selfcol && pool_cap > 0) { cmask[tt] = 0; cmask[tt + T] = 0; }
            __syncthreads();
            continue;
        }
        // ---- voxel phase, my two voxels one after the other
        double vel2 = 0;
        VoxState S0, S1;
#pragma nounroll
        for (int h = 0; h < 2; ++h) {
            if (h ? valid1 : valid0) {
                const int l = tt + h * T;
                const DVoxClass& C = vct[h ? cls1 : cls0];
                VoxState S;
                d3 F = mk3(acc[l], acc[NV + l], acc[2 * NV + l]), M = mk3(acc[3 * NV + l], acc[4 * NV + l], acc[5 * NV + l]);
                S.pos = mk3(ps[l], ps[NV + l], ps[2 * NV + l]); S.scale = ps[3 * NV + l];
                S.ang = mkq(ps[4 * NV + l], ps[5 * NV + l], ps[6 * NV + l], ps[7 * NV + l]);
                S.lm = h ? lm1 : lm0; S.am = h ? am1 : am0;
                const d3 vel = S.lm * C.mass_inv;
                F = F + (vel * (-R.slow_z)) * C.c_lin;
                const int rowd = h ? rowd1 : rowd0;
                if (rowd != 0) F = fused_contact_forces<NV>(B, R, ps, F, S.pos, S.scale, l, base + l, rowd, cmask, rc_code, rc_a1);
                const double w2 = voxel_update(B, R, C, base + l, fetch, K.time, K.act_sin, K.act_cos, K.prenatal_c, F, M, vel, S, -1, 0, false, mk3(0, 0, 0),
                                               h ? phs1 : phs0, h ? phc1 : phc0, h ? amp1 : amp0);
                vel2 = w2 > vel2 ? w2 : vel2;
                if (h) { lm1 = S.lm; am1 = S.am; S1 = S; } else { lm0 = S.lm; am0 = S.am; S0 = S; }
#pragma unroll
                for (int k = 0; k < 6; ++k) acc[k * NV + l] = 0.0;
            }
        }
        if (ctl_thread) fused_control_begin(R, rs, step_cap, it + 1 < iters, Knext);
        if (selfcol) {                         // SS.MaxVoxVel for the collision horizon (VX_Sim.cpp:1625-1649)
            vel2 = wave_max_nonneg(vel2);
            if (lane == 0) atomicMax(&rs.maxvel2_bits, (unsigned long long)__double_as_longlong(vel2));
        }
        VXH_T_MARK(4)
        __syncthreads();                       // (C) every read of the old poses is done
        if (valid0) {
            const int l = tt;
            ps[l] = S0.pos.x; ps[NV + l] = S0.pos.y; ps[2 * NV + l] = S0.pos.z; ps[3 * NV + l] = S0.scale;
            ps[4 * NV + l] = S0.ang.w; ps[5 * NV + l] = S0.ang.x; ps[6 * NV + l] = S0.ang.y; ps[7 * NV + l] = S0.ang.z;
        }
        if (valid1) {
            const int l = tt + T;
            ps[l] = S1.pos.x; ps[NV + l] = S1.pos.y; ps[2 * NV + l] = S1.pos.z; ps[3 * NV + l] = S1.scale;
            ps[4 * NV + l] = S1.ang.w; ps[5 * NV + l] = S1.ang.x; ps[6 * NV + l] = S1.ang.y; ps[7 * NV + l] = S1.ang.z;
        }
        VXH_T_MARK(5)
        if (ctl_thread) { fused_control_horizon(R, rs, Knext); s_div = 0; }
        __syncthreads();                       // (A) control + every voxel's published pose visible
        VXH_T_MARK(0)
    }
    VXH_T_FLUSH
    // ---- back to HBM
    {
        const int b1 = rs.steps & 1;
        if (valid0) {
            const int v = base + tid, l = tid;
            POS(b1, 0, v) = ps[l]; POS(b1, 1, v) = ps[NV + l]; POS(b1, 2, v) = ps[2 * NV + l]; SCALE(b1, v) = ps[3 * NV + l];
            QUAT(0, v) = ps[4 * NV + l]; QUAT(1, v) = ps[5 * NV + l]; QUAT(2, v) = ps[6 * NV + l]; QUAT(3, v) = ps[7 * NV + l];
            LINMOM(0, v) = lm0.x; LINMOM(1, v) = lm0.y; LINMOM(2, v) = lm0.z; ANGMOM(0, v) = am0.x; ANGMOM(1, v) = am0.y; ANGMOM(2, v) = am0.z;
        }
        if (valid1) {
            const int v = base + tid + T, l = tid + T;
            POS(b1, 0, v) = ps[l]; POS(b1, 1, v) = ps[NV + l]; POS(b1, 2, v) = ps[2 * NV + l]; SCALE(b1, v) = ps[3 * NV + l];
            QUAT(0, v) = ps[4 * NV + l]; QUAT(1, v) = ps[5 * NV + l]; QUAT(2, v) = ps[6 * NV + l]; QUAT(3, v) = ps[7 * NV + l];
            LINMOM(0, v) = lm1.x; LINMOM(1, v) = lm1.y; LINMOM(2, v) = lm1.z; ANGMOM(0, v) = am1.x; ANGMOM(1, v) = am1.y; ANGMOM(2, v) = am1.z;
        }
    }
    if (eX0 != -1) B.small_angle[base + (eX0 & 1023)] = (unsigned char)(mb0 & 3u);
    if (eX1 != -1) B.small_angle[base + (eX1 & 1023)] = (unsigned char)(mb1 & 3u);
    if (eY0 != -1) B.small_angle[nv + (base + (eY0 & 1023))] = (unsigned char)((mb0 >> 2) & 3u);
    if (eY1 != -1) B.small_angle[nv + (base + (eY1 & 1023))] = (unsigned char)((mb1 >> 2) & 3u);
    if (eZ0 != -1) B.small_angle[2u * nv + (base + (eZ0 & 1023))] = (unsigned char)((mb0 >> 4) & 3u);
    if (eZ1 != -1) B.small_angle[2u * nv + (base + (eZ1 & 1023))] = (unsigned char)((mb1 >> 4) & 3u);
    if (tid == 0) { B.rstate[r] = rs; if (B.rstate_mirror) B.rstate_mirror[r] = rs; }
}

}  // namespace vxh
