// HIP kernels of the batched Voxelyze time-stepper (gfx950).  FP64 throughout, SoA state in HBM
// (device_types.hpp).  Reference loops being replaced (paths under evosoro/_voxcad/Voxelyze/):
//   step_control    CVX_Sim::TimeStep prologue + StopConditionMet + UpdateCollisions and the `CurTime += dt`
//                   epilogue (VX_Sim.cpp:1054-1135,1398-1423,1729-1755,1929)
//   rebuild_rows    CVX_Sim::CalcL1Bonds (VX_Sim.cpp:2357-2413)
//   bond_compute    CVXS_BondInternal::CalcLinForce / UpdateBondStrain / AddDampForces (VXS_BondInternal.cpp:56-346)
//   voxel_update    CVXS_Voxel::EulerStep / CalcTotalForce / CalcTotalMoment / CalcFloorEffect (VXS_Voxel.cpp:169-758),
//                   CVXS_BondCollision::CalcContactForce (VXS_BondCollision.cpp:41-59), MaxVoxVel of UpdateStats
// Two launch shapes share those device functions:
//   k_robot_steps<BLOCK,NACC,MESH,TABG>  fused path (kernels_fused.hpp): ONE workgroup per robot (robots up to BLOCK voxels),
//                         the robot RESIDENT in the CU for a whole launch of many time steps: voxel momenta in registers,
//                         poses and force accumulators in LDS, bonds taken from per-axis compacted lists.  Per step only
//                         the bond history crosses L2/HBM.
//   k_step_begin / k_bonds / k_voxels   streaming path for lattices of any size (one thread per bond slot / voxel,
//                         state and bond outputs through HBM).
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
// qmul(a, f) for a quaternion whose x component is an exact zero (what FromAngleToPosX returns): the four products with it and their
// additions left out -- x +- 0 == x, so the same values; the compiler may not drop a multiplication by a zero it cannot prove finite
__device__ __forceinline__ dq qmul_x0(dq a, dq f)
{
    return mkq(a.w * f.w - a.y * f.y - a.z * f.z,
               a.w * f.x + a.y * f.z - a.z * f.y,
               a.w * f.y + a.y * f.w + a.z * f.x,
               a.w * f.z - a.y * f.x + a.z * f.w);
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
// v -> conj(q) v q as a matrix-vector product: 28 operations once, 9 per vector (rotinv: 24 each).  NOT for unit quaternions only:
// the diagonal carries |q|^2, as the sandwich does.  The bond frame's quaternion is not always a unit one -- FromAngleToPosX's
// small-angle branch returns (1 - (y^2 + z^2) / 2, 0, y, z), of norm^2 1 + (y^2 + z^2)^2 / 4, up to 1 + 5e-9 -- and the reference's
// RotateVec3DInv scales the back-rotated forces and moments of such a bond by it.  Until round 3 the diagonal was 1 - 2 (y^2 + z^2):
// every large-angle bond with a small bend differed from the reference by that factor, 1e-13 .. 1e-9 of its force, step after step
// in the same direction -- the "creep" of the long runs (scripts/dev_gpu_diag.py drift7: one step of engine and oracle from the
// same state differed by 1e-12 voxel, 500 x what a one-ulp change of the inputs does to the oracle).
struct RotInv {
    double m00, m01, m02, m10, m11, m12, m20, m21, m22;
    __device__ __forceinline__ explicit RotInv(dq q)
    {
        const double x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
        const double xx = x2 * q.x, yy = y2 * q.y, zz = z2 * q.z, xy = x2 * q.y, xz = x2 * q.z, yz = y2 * q.z, wx = x2 * q.w, wy = y2 * q.w, wz = z2 * q.w;
        const double n = (q.w * q.w + q.x * q.x) + (q.y * q.y + q.z * q.z);
        m00 = n - (yy + zz); m11 = n - (xx + zz); m22 = n - (xx + yy);
        m01 = xy + wz; m10 = xy - wz;        // row i of R^T = column i of the rotation matrix of q
        m02 = xz - wy; m20 = xz + wy;
        m12 = yz + wx; m21 = yz - wx;
    }
    __device__ __forceinline__ d3 operator()(d3 f) const
    {
        return mk3(m00 * f.x + m01 * f.y + m02 * f.z, m10 * f.x + m11 * f.y + m12 * f.z, m20 * f.x + m21 * f.y + m22 * f.z);
    }
};

// ToXDirBond / ToOrigDirBond, VX_Bond.h:45-48 ; axis 0 = X, 1 = Y, 2 = Z.  The axis is a template argument: the frame
// change is a compile-time permutation, not a chain of selects.
template <int A> __device__ __forceinline__ d3 to_xdir(d3 p)
{
    if constexpr (A == 1) return mk3(p.y, -p.x, p.z);
    else if constexpr (A == 2) return mk3(p.z, p.y, -p.x);
    else return p;
}
template <int A> __device__ __forceinline__ dq to_xdir(dq q)
{
    if constexpr (A == 1) return mkq(q.w, q.y, -q.x, q.z);
    else if constexpr (A == 2) return mkq(q.w, q.z, q.y, -q.x);
    else return q;
}
template <int A> __device__ __forceinline__ d3 to_orig(d3 p)
{
    if constexpr (A == 1) return mk3(-p.y, p.x, p.z);
    else if constexpr (A == 2) return mk3(-p.z, p.y, p.x);
    else return p;
}


// FP64 sqrt / divide / reciprocal as the instruction sequences the device library emits for them, minus the operand
// range scaling (v_div_scale / v_ldexp) and the special-value fix-up (v_div_fixup, v_cmp_class): for finite, normal,
// non-zero operands -- everything on this path, which works in metres, newtons and unit quaternions -- the results are
// bit-identical to sqrt(), a / b and 1.0 / b at 10, 8 and 7 instead of 18, 11 and 11 instructions.  The *_nn variant
// also returns 0 for a zero argument.
__device__ __forceinline__ double vsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
// 1 / sqrt(x), x > 0 normal: v_rsq_f64 + ONE third-order step, <= 1 ulp.  With e = 1 - x y0^2 the exact value is y0 / sqrt(1 - e) =
// y0 (1 + e/2 + 3 e^2/8 + 5 e^3/16 + ...); v_rsq_f64 is good to 2^-23 (|e| <= 2^-22), so the series cut after e^2 is off by 2^-68
// and what is left is rounding: half an ulp from the product x y0 inside e, half from the final fused multiply-add.  (Until round
// 4 a second, second-order step followed: four more instructions per call, five to six calls per large-angle bond, for the last
// quarter of an ulp.)
__device__ __forceinline__ double vrsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-x * y, y, 1.0);
    return __builtin_fma(y * e, __builtin_fma(e, 0.375, 0.5), y);      // y (1 + e/2 + 3 e^2/8)
}
__device__ __forceinline__ double vsqrt_nn(double x) { const double s = vsqrt(x); return x == 0 ? 0.0 : s; }
__device__ __forceinline__ double vrcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    return __builtin_fma(e, r, r);
}
__device__ __forceinline__ double vdiv(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
    double q = a * r;
    e = __builtin_fma(-b, q, a);
    return __builtin_fma(e, r, q);
}

// acos for |x| <= 1, <= 1 ulp: the usual reduction (|x| < 0.5: pi/2 - asin x; else 2 asin sqrt((1 - |x|)/2)) with
// asin(s) = s + s z R(z), z = s^2 in [0, 1/4], R a degree-11 polynomial (Chebyshev-node interpolant of
// (asin(sqrt z) - sqrt z)/(z sqrt z), relative error 6e-17).  The coefficients are pinned to scalar registers: as
// literals the compiler moves each of them into a vector register pair per evaluation (FP64 has no 64-bit literals).
__device__ __forceinline__ double sconst(double c) { asm volatile("" : "+s"(c)); return c; }
// z P(z) of asin(s) = s (1 + z P(z)), z = s^2 <= 1/4 (the device library's acos polynomial)
__device__ __forceinline__ double asin_zp(double z)
{
    double p = sconst(0.028169218060881414);
    p = __builtin_fma(p, z, sconst(-0.010749050339697808));
    p = __builtin_fma(p, z, sconst(0.01603551434914882));
    p = __builtin_fma(p, z, sconst(0.0078029494773533175));
    p = __builtin_fma(p, z, sconst(0.011875494382636922));
    p = __builtin_fma(p, z, sconst(0.013929652902326633));
    p = __builtin_fma(p, z, sconst(0.017355259955786323));
    p = __builtin_fma(p, z, sconst(0.02237204763174451));
    p = __builtin_fma(p, z, sconst(0.03038194736709848));
    p = __builtin_fma(p, z, sconst(0.044642857103423646));
    p = __builtin_fma(p, z, sconst(0.07500000000020764));
    p = __builtin_fma(p, z, sconst(0.1666666666666665));
    p *= z;
    return p;
}
__device__ __forceinline__ double vacos(double x)
{
    const double ax = fabs(x);
    const bool big = ax >= 0.5;
    const double z = big ? __builtin_fma(ax, -0.5, 0.5) : x * x;
    const double p = asin_zp(z);
    if (big) {
        const double s = vsqrt_nn(z);
        const double r = 2.0 * __builtin_fma(s, p, s);
        return x > 0 ? r : 3.141592653589793 - (r - 1.2246467991473532e-16);
    }
    return 1.5707963267948966 - (x + __builtin_fma(x, p, -6.123233995736766e-17));
}

// max of a non-negative double over the 64 lanes of a (fully active) wavefront, returned in every lane.  Non-negative doubles order
// like their bit patterns read as unsigned 64-bit integers, i.e. lexicographically by (high word, low word): the maximum of the
// high words first (v_max_u32 takes a DPP source: ONE instruction per stage -- four row shifts, two row broadcasts), then the
// maximum of the low words among the lanes that hold that high word.  14 vector instructions where the FP64 compare-and-select
// version (v_max_f64 is VOP3: no DPP) took 36; same bits.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax_stage(unsigned v)
{
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);     // 0 where no lane feeds this one
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_umax(unsigned v)
{
    v = dpp_umax_stage<0x111, 0xf>(v);      // row_shr:1
    v = dpp_umax_stage<0x112, 0xf>(v);      // row_shr:2
    v = dpp_umax_stage<0x114, 0xf>(v);      // row_shr:4
    v = dpp_umax_stage<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row of 16 holds the row's maximum
    v = dpp_umax_stage<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_umax_stage<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the maximum
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ double wave_max_nonneg(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned hi = wave_umax((unsigned)(b >> 32));
    const unsigned lo = wave_umax((unsigned)(b >> 32) == hi ? (unsigned)b : 0u);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// min / max of any doubles over the 64 lanes of a (fully active) wavefront, returned in every lane (a lane without a source keeps its
// own value, so no sign convention is needed)
template <int CTRL, int ROW_MASK, bool MAX>
__device__ __forceinline__ double dpp_minmax_stage(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)b, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(b >> 32), (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    const double o = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    return MAX ? (o > v ? o : v) : (o < v ? o : v);
}
template <bool MAX>
__device__ __forceinline__ double wave_minmax(double v)
{
    v = dpp_minmax_stage<0x111, 0xf, MAX>(v);
    v = dpp_minmax_stage<0x112, 0xf, MAX>(v);
    v = dpp_minmax_stage<0x114, 0xf, MAX>(v);
    v = dpp_minmax_stage<0x118, 0xf, MAX>(v);
    v = dpp_minmax_stage<0x142, 0xa, MAX>(v);
    v = dpp_minmax_stage<0x143, 0xc, MAX>(v);
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

#define VXH_PI 3.14159265358979
#define VXH_DISCARD_ANGLE_RAD 1e-7
#define VXH_SMALL_ANGLE_RAD 1.732e-2
#define VXH_SA_BOND_BEND_RAD 0.05
#define VXH_SA_BOND_EXT_PERC 1.30
#define VXH_HYST 1.1

// CQuat::FromAngleToPosX, Vec3D.h:208-237: the rotation that takes `from` onto +X, axis (0, n.z, -n.y)/sin(theta), angle
// theta = acos(n.x).  The reference evaluates (cos, sin)(theta/2) through acos + sincos; here the same quaternion is
// written with the half-angle identities cos(theta/2) = sqrt((1 + n.x)/2), sin(theta/2)/sin(theta) = 1/(2 cos(theta/2)):
//     (c, 0, n.z/(2c), -n.y/(2c)),  c = sqrt((1 + n.x)/2)
// acos is ill-conditioned near n.x = 1 (d theta = d n.x / sin theta), so the reference's own value moves by up to 1e-12
// (relative) under a 1-ulp change of `from`; the identity form agrees with it to 8e-13 max / 4e-15 median over bend angles
// of 1..35 degrees (DESIGN.md "Numerics") at a quarter of the instructions.  The theta > PI - 1e-7 cut is kept on n.x.
// Round 4 (instruction diet): the function also returns |from| -- the large-angle branch of CalcLinForce needs the bond's length
// (Pos2.x = |x2 - x1| - NomDistance, VXS_BondInternal.cpp:104) and this normalisation needs 1 / |from|: one refined v_rsq_f64 gives
// both (len = L2 * rsqrt(L2), <= 1.5 ulp, where the reference takes a correctly rounded sqrt of the un-rotated difference: a relative
// 2e-16 on a length whose elongation is 1e-4 .. 1e-1 of it); and the SmallAngle test |y/x|, |z/x| < SMALL_ANGLE_RAD is made on the
// cross-multiplied form |y|, |z| < SMALL_ANGLE_RAD |x| (x == 0: false, like the reference's inf / NaN quotients), so that the
// reciprocal of x is only formed for the lanes that take that branch.
__device__ __forceinline__ dq from_angle_to_pos_x(d3 from, double& len)
{
    const double L2 = from.x * from.x + from.y * from.y + from.z * from.z;
    if (from.x == 0 && from.y == 0 && from.z == 0) { len = 0.0; return mkq(1, 0, 0, 0); }
    const double li = vrsqrt(L2);             // NormalizeFast: from * (1 / |from|)
    len = L2 * li;
    const double ax = VXH_SMALL_ANGLE_RAD * fabs(from.x);
    if (fabs(from.y) < ax && fabs(from.z) < ax) {
        const double rx = vrcp(from.x);
        const double y = 0.5 * (from.z * rx), z = -0.5 * (from.y * rx);
        return mkq(1 + 0.5 * (-y * y - z * z), 0, y, z);
    }
    const d3 n = from * li;
    if (n.x < -0.999999999999995) return mkq(0, 0, 1, 0);     // cos(PI - DISCARD_ANGLE_RAD)
    const double s = 0.5 + 0.5 * n.x, ri = vrsqrt(s);         // c = sqrt(s), 1 / (2 c) = ri / 2
    const double c = s * ri, h = 0.5 * ri;
    return mkq(c, 0, n.z * h, -n.y * h);
}
// CQuat::ToRotationVector, Vec3D.h:270-285
// 1 - w * w with the product rounded on its own, as the reference's FMA-less build forms it.  For a small rotation (w within 1e-4
// of 1) the rounding of w * w is a relative error of 1e-12 in the difference; contracted into one fused multiply-add the engine's
// value was the more accurate one -- and 1e-12 away from the reference's, which showed as a jump of the error against the oracle
// from 1e-14 to 6e-12 (relative, angular velocity) on the step the first bonds of a violently starting robot turn large-angle
// (tests/golden/vxa/lw_hexapus.vxa, step 4; scripts/dev_gpu_diag.py drift3, round 3).
#pragma clang fp contract(off)
__device__ __forceinline__ double one_minus_square(double w) { const double ww = w * w; return 1.0 - ww; }
#pragma clang fp contract(fast)
// Round 4 (instruction diet).  f = 2 * (angle / sin(angle/2) / 2) in three branches:
//   sl < SLTHRESH   the reference's sqrt((2 - 2w) / sl) as (2 - 2w) * rsqrt((2 - 2w) * sl), sl = 1 - fl(w w) with the reference's own
//                   rounding of w w (one_minus_square above: for a small rotation that rounding is a relative 1e-12 of sl, and the
//                   engine must make the same "error");
//   else, w >= 0.5  (every bond that is not folded back on itself)  acos(w) = 2 asin(s), s = sqrt(z), z = (1 - w)/2, asin(s) = s (1 + z P(z))
//                   with vacos' polynomial; and sqrt(sl) = sqrt((1 - w)(1 + w)) = s sqrt(2 + 2w), so that
//                       acos(w) / sqrt(sl) = (2 + 2 z P(z)) * rsqrt(2 + 2w)
//                   -- ONE refined v_rsq_f64 of a well-conditioned argument where acos-then-divide takes a sqrt and a rsqrt (-14 vector
//                   instructions per call, two calls per large-angle bond).  It does not go through sl, hence not through the
//                   rounding of w w: on this branch sl > 2.4e-3, that rounding is a relative < 2.5e-14 of the result.
//   else            acos(w) * rsqrt(sl) as before (never taken by a sane bond; kept for the function's contract).
// The factor 2 of the rotation vector is folded into f (exact).
// SEL (the resident kernel): the select form -- numerator and rsqrt argument of the lane's branch chosen first, ONE refined v_rsq_f64
// for both (the same operations on the same values per lane as the branched form: same bits, scripts/dev_gpu_diag.py statehash).
// Measured, same box: resident kernel 25.15 -> 24.96 us per step (its wavefronts are of both kinds, and run both branches); the wide
// kernel the other way, 64 x 6^3 5.6 -> 5.7, swimmers 13.8 -> 14.0, 512 x 8^3 16.75 -> 17.05 -- hence a switch.
template <bool SEL = false>
__device__ __forceinline__ double rotvec_factor(double w, double slthresh)
{
    const double sl = one_minus_square(w);
    if constexpr (SEL) {
        if (__builtin_expect(w < 0.5 && sl >= slthresh, 0)) return 2.0 * vacos(w) * vrsqrt(sl);
        const double z = __builtin_fma(w, -0.5, 0.5);
        const double p = asin_zp(z);
        const bool approx = sl < slthresh;
        const double a = __builtin_fma(w, -2.0, 2.0), a2 = __builtin_fma(w, -4.0, 4.0);
        const double num = approx ? a2 : __builtin_fma(p, 4.0, 4.0);
        const double arg = approx ? a * sl : __builtin_fma(w, 2.0, 2.0);
        const double f = num * vrsqrt(arg);
        return sl <= 0 ? 0.0 : f;
    }
    if (sl <= 0) return 0.0;                   // (sl > 0 from here: |w| < 1, the reference's clamp of w to 1 cannot act)
    if (sl < slthresh) {
        const double a = __builtin_fma(w, -2.0, 2.0), a2 = __builtin_fma(w, -4.0, 4.0);
        return a2 * vrsqrt(a * sl);
    }
    if (w >= 0.5) {
        const double z = __builtin_fma(w, -0.5, 0.5);
        const double p = asin_zp(z);
        return __builtin_fma(p, 4.0, 4.0) * vrsqrt(__builtin_fma(w, 2.0, 2.0));
    }
    return 2.0 * vacos(w) * vrsqrt(sl);
}
template <bool SEL = false>
__device__ __forceinline__ d3 to_rotvec(dq q, double slthresh)
{
    const double f = rotvec_factor<SEL>(q.w, slthresh);
    return mk3(q.x * f, q.y * f, q.z * f);
}

// Contact rows (DBatch::col_partner / col_a1 / col_code): robot r owns the block [col_begin, col_begin + col_cap * nsurf), entry k of
// the row of its surface voxel i at col_begin + k * nsurf + i (partner-major inside the block: coalesced across a wavefront).
// `row` = the batch-wide row number surf_begin + i, as DBatch::col_cnt is indexed.
__device__ __forceinline__ size_t col_at(const DRobot& R, int k, int row)
{
    return (size_t)R.col_begin + (size_t)k * (size_t)R.nsurf + (size_t)(row - R.surf_begin);
}

__device__ __forceinline__ int robot_of(const DBatch& B, int vslot)
{
    return __builtin_amdgcn_readfirstlane(B.wave_robot[vslot >> 6]);
}

struct BondOut { d3 f1, m1, f2, m2; double strain1, strain2; bool diverged; };
// History of one bond in the 6-double layout of DBatch::hist + its flag bits (bit 0 SmallAngle, bit 1 large layout).
// Loaded by the caller BEFORE the arithmetic and stored after it, so the memory operations of a bond are issued
// together instead of being scattered through the math.
struct BondHist { double p0, p1, p2, g0, g1, g2; unsigned flags; bool store_hist; };

__device__ __forceinline__ BondHist load_bond_hist(const DBatch& B, int slot)
{
    BondHist h;
    h.p0 = HIST(0, slot); h.p1 = HIST(1, slot); h.p2 = HIST(2, slot);
    h.g0 = HIST(3, slot); h.g1 = HIST(4, slot); h.g2 = HIST(5, slot);
    h.flags = B.small_angle[slot];
    h.store_hist = false;
    return h;
}
__device__ __forceinline__ void store_bond_hist(const DBatch& B, int slot, const BondHist& h, unsigned old_flags)
{
    if (h.store_hist) {
        HIST(0, slot) = h.p0; HIST(1, slot) = h.p1; HIST(2, slot) = h.p2;
        HIST(3, slot) = h.g0; HIST(4, slot) = h.g1; HIST(5, slot) = h.g2;
    }
    if (h.flags != old_flags) B.small_angle[slot] = (unsigned char)h.flags;
}

// One internal bond in the BOND frame (the bond lies along +x: the caller has applied ToXDirBond to the relative position and
// to both orientations): pure arithmetic, `H` in/out.  The outputs are still in the permuted global frame (the caller applies
// ToOrigDirBond); f2 is only computed for heterogeneous bonds (F2 = -F1 is enforced after the back-rotation otherwise).
// damp_on = a previous step exists (no damping on the first one, dt == 0 then: VXS_BondInternal.cpp:311).
template <bool SEL = false>
__device__ __forceinline__ BondOut bond_compute_xframe(const DBatch& B, const DBondClass& C, BondHist& H,
                                                       d3 xrel, dq a1, dq a2, double nom_dist, bool damp_on)
{
    BondOut o;
    d3 rel = rotinv(a1, xrel);
    // Angle2 in voxel 1's frame, conj(a1) a2 (NewAng2): its w now, for the mode test; the vector part only where it is used, in the
    // small-angle branch (a wavefront of large-angle bonds skips those 12 operations)
    const double new2w = a1.w * a2.w + a1.x * a2.x + a1.y * a2.y + a1.z * a2.z;

    // small/large-angle switch with hysteresis (VXS_BondInternal.cpp:72-77).  SmallTurn = (|z|+|y|)/x and
    // ExtendPerc = x/NomDistance are only ever compared with constants, so the comparisons are made on the
    // cross-multiplied form (no division); the sign cases keep the IEEE outcome of the quotient (x < 0: quotient <= 0,
    // x == 0: +inf or NaN).
    // (Round 6: written as lane masks -- bitwise operators on the comparisons' results -- instead of nested ifs: the compiler had turned
    // every `if` into a save-exec / branch pair, a dozen scalar round trips at the head of a chain that issues one instruction per ~7
    // cycles; the same truth table, NaN cases included: rel.x unordered -> neither xpos nor xneg -> the reference's `else` arm.)
    bool small = (H.flags & 1u) != 0, changed = false;
    {
        const double t = fabs(rel.z) + fabs(rel.y);
        const bool xpos = rel.x > 0, xneg = rel.x < 0;
        const bool turn_lt = (xpos & (t < VXH_SA_BOND_BEND_RAD * rel.x)) | xneg;                                  // SmallTurn < BEND
        const bool turn_gt = (xpos & (t > (VXH_HYST * VXH_SA_BOND_BEND_RAD) * rel.x)) | (!xpos & !xneg & (t > 0));   // SmallTurn > HYST * BEND
        const bool ext_lt = rel.x < VXH_SA_BOND_EXT_PERC * nom_dist, ext_gt = rel.x > (VXH_HYST * VXH_SA_BOND_EXT_PERC) * nom_dist;
        const bool to_small = !small & (new2w > B.small_angle_w) & turn_lt & ext_lt;
        const bool to_large = small & (!(new2w > B.smallish_angle_w) | turn_gt | ext_gt);
        changed = to_small | to_large;
        small = (small | to_small) & !to_large;
    }
    // (Round 4, measured: a wavefront whose 64 bonds are of both modes runs both branches below, and in the bench population 6 % of the
    // bonds are small-angle, so nearly every wavefront does.  The ceiling of sorting the bonds by mode -- every bond forced large-angle, the
    // small branch compiled out, timing only -- is 25.15 -> 24.92 us per step: not worth a per-launch re-sort of DBatch::bsched.)

    // Angle1 is (0, y, z) in both modes: zero in the small one, and the rotation vector of FromAngleToPosX's quaternion, whose x is
    // an exact zero, in the large one -- the x component is left out of everything below (0 - a == -a, a - 0 == a: the same bits)
    d3 pos2;
    double ang1y, ang1z;
    dq rot, qb2;                               // qb2: Angle2 in the bond frame
    if (small) {
        ang1y = 0; ang1z = 0;
        qb2 = mkq(new2w, a1.w * a2.x - a1.x * a2.w - a1.y * a2.z + a1.z * a2.y,
                         a1.w * a2.y + a1.x * a2.z - a1.y * a2.w - a1.z * a2.x,
                         a1.w * a2.z - a1.x * a2.y + a1.y * a2.x - a1.z * a2.w);      // qmul(conj(a1), a2)
        pos2 = mk3(rel.x - nom_dist, rel.y, rel.z);
        rot = conj(a1);
    } else {
        double len;
        const dq align = from_angle_to_pos_x(rel, len);
        rot = qmul_x0(align, conj(a1));
        pos2 = mk3(len - nom_dist, 0, 0);
        const double f1 = rotvec_factor<SEL>(align.w, B.slthresh_acos2sqrt);
        ang1y = align.y * f1; ang1z = align.z * f1;
        qb2 = qmul(rot, a2);               // (re-associated as align (conj(a1) a2): needs all of conj(a1) a2, 12 operations more than it saves)
    }
    const d3 ang2 = to_rotvec<SEL>(qb2, B.slthresh_acos2sqrt);   // one instance for both modes: a mixed wave runs it once

    // axial stress (UpdateBondStrain, VXS_BondInternal.cpp:189-307; linear materials): the reference's series-spring
    // iteration is linear in the strain = elongation / L, its three factors (with the 1 / L) are constants of the bond
    // class (model.cpp make_bond_class, DBondClass)
    o.strain1 = C.strain_a1_L * pos2.x;        // CurStrainV1 / CurStrainV2 (SetStrainDir), read by the land_water surface mesh
    o.strain2 = C.strain_a2_L * pos2.x;
    o.diverged = pos2.x > C.L100;              // strain > 100, VX_Sim.cpp:1775

    // beam equations (VXS_BondInternal.cpp:128-153)
    d3 f1 = mk3(C.kf_L * pos2.x, C.b1 * pos2.y - C.b2 * (ang1z + ang2.z), C.b1 * pos2.z + C.b2 * (ang1y + ang2.y));
    d3 f2 = -f1;
    const double tors = C.a2 * ang2.x;         // a2 (Angle2.x - Angle1.x); Moment1.x is its negative
    d3 m1 = mk3(-tors, C.b2 * pos2.z + C.b3 * (2 * ang1y + ang2.y), -C.b2 * pos2.y + C.b3 * (2 * ang1z + ang2.z));
    d3 m2 = mk3(tors, C.b2 * pos2.z + C.b3 * (ang1y + 2 * ang2.y), -C.b2 * pos2.y + C.b3 * (ang1z + 2 * ang2.z));

    // velocity damping from the finite-differenced bond-frame pose (AddDampForces :310-346); skipped on the step the
    // mode flips, and the history is only refreshed when it runs
    if (!changed) {
        if (damp_on) {
            const bool hl = (H.flags & 2u) != 0;        // expand the stored history (see DBatch::hist)
            const d3 hpos2 = mk3(H.p0, hl ? 0.0 : H.p1, hl ? 0.0 : H.p2);
            const double hang1y = hl ? H.p1 : 0.0, hang1z = hl ? H.p2 : 0.0;
            const d3 hang2 = mk3(H.g0, H.g1, H.g2);
            // differences of the bond-frame pose; 1/dt and BondDampingZ/2 are inside the d* constants (DBondClass)
            const d3 v = pos2 - hpos2, w2 = ang2 - hang2;
            const double w1y = ang1y - hang1y, w1z = ang1z - hang1z;      // (w1.x == 0)
            f1 = f1 + mk3(C.dA1 * v.x, C.dB1 * v.y - C.dF1 * (w1z + w2.z), C.dB1 * v.z + C.dF1 * (w1y + w2.y));
            if (!C.homogeneous)
                f2 = f2 + mk3(-C.dA2 * v.x, -C.dB2 * v.y + C.dF2 * (w1z + w2.z), -C.dB2 * v.z - C.dF2 * (w1y + w2.y));
            m1 = m1 + mk3(-C.dT1 * w2.x, C.dG1 * v.z + C.dH1 * (2 * w1y + w2.y), -C.dG1 * v.y + C.dH1 * (2 * w1z + w2.z));
            m2 = m2 + mk3(C.dT2 * w2.x, C.dG2 * v.z + C.dH2 * (w1y + 2 * w2.y), -C.dG2 * v.y + C.dH2 * (w1z + 2 * w2.z));
        }
        // _LastPos2 / _LastAngle1 / _LastAngle2 in the layout of the mode that produced them (ang1 == 0 in small mode;
        // pos2.y == pos2.z == ang1.x == 0 in large mode)
        H.p0 = pos2.x; H.p1 = small ? pos2.y : ang1y; H.p2 = small ? pos2.z : ang1z;
        H.g0 = ang2.x; H.g1 = ang2.y; H.g2 = ang2.z;
        H.flags = (small ? 1u : 2u);
        H.store_hist = true;
    } else {
        H.flags = (H.flags & 2u) | (small ? 1u : 0u);
    }

    // back to the global frame (:158-171): three or four vectors go through the same RotateVec3DInv(rot, .), so the
    // rotation is expanded once into its matrix (conj(q) v q = R^T v for a unit quaternion; |rot| = 1 to rounding)
    const RotInv T(rot);
    o.f1 = T(f1);
    if (!C.homogeneous) o.f2 = T(f2);
    o.m1 = T(m1);
    o.m2 = T(m2);
    return o;
}

// ... along axis A between voxel 1 (negative side) and voxel 2, the axis a compile-time constant (fused and streaming kernels)
template <int A, bool SEL = false>
__device__ __forceinline__ BondOut bond_compute(const DBatch& B, const DBondClass& C, BondHist& H,
                                                d3 p1, dq q1, double s1, d3 p2, dq q2, double s2,
                                                bool damp_on)
{
    BondOut o = bond_compute_xframe<SEL>(B, C, H, to_xdir<A>(p2 - p1), to_xdir<A>(q1), to_xdir<A>(q2), (s1 + s2) * 0.5, damp_on);
    o.f1 = to_orig<A>(o.f1);
    o.f2 = C.homogeneous ? -o.f1 : to_orig<A>(o.f2);
    o.m1 = to_orig<A>(o.m1);
    o.m2 = to_orig<A>(o.m2);
    return o;
}

// ... the axis a per-lane run-time value (tiled kernel: one flat bond list per tile, all axes in one round).  The same
// permutations as to_xdir<A> / to_orig<A> written with selects: bit-identical results (negation is exact).
__device__ __forceinline__ d3 to_xdir_rt(int a, d3 p) { return mk3(a == 0 ? p.x : (a == 1 ? p.y : p.z), a == 1 ? -p.x : p.y, a == 2 ? -p.x : p.z); }
__device__ __forceinline__ dq to_xdir_rt(int a, dq q) { return mkq(q.w, a == 0 ? q.x : (a == 1 ? q.y : q.z), a == 1 ? -q.x : q.y, a == 2 ? -q.x : q.z); }
__device__ __forceinline__ d3 to_orig_rt(int a, d3 p) { return mk3(a == 0 ? p.x : (a == 1 ? -p.y : -p.z), a == 1 ? p.x : p.y, a == 2 ? p.x : p.z); }
__device__ __forceinline__ BondOut bond_compute_rt(int axis, const DBatch& B, const DBondClass& C, BondHist& H,
                                                   d3 p1, dq q1, double s1, d3 p2, dq q2, double s2, bool damp_on)
{
    BondOut o = bond_compute_xframe(B, C, H, to_xdir_rt(axis, p2 - p1), to_xdir_rt(axis, q1), to_xdir_rt(axis, q2), (s1 + s2) * 0.5, damp_on);
    o.f1 = to_orig_rt(axis, o.f1);
    o.f2 = C.homogeneous ? -o.f1 : to_orig_rt(axis, o.f2);
    o.m1 = to_orig_rt(axis, o.m1);
    o.m2 = to_orig_rt(axis, o.m2);
    return o;
}

// ---- land_water surface mesh and fluid drag (LW/VX_Sim.cpp:1516-1597), arithmetic shared by the resident kernel
// (kernels_fused.hpp: fused_drag) and the streaming kernels (k_mesh_vertices / k_facets below).  Every deformable surface
// vertex = mean over the <= 7 voxels touching that lattice corner of Pos + R(Angle) * corner offset, corner offsets from the
// bond strains of the PREVIOUS step (CornerPosCur/CornerNegCur, LW/VXS_Voxel.cpp:472-475; GetCurVLoc
// LW/VX_MeshUtil.cpp:388-428); every voxel sums the quadratic drag of the two triangles on each of its exposed faces, in the
// reference's facet order.
__device__ __forceinline__ d3 cross3(d3 a, d3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ double dot3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 normalized3(d3 a) { const double l = vsqrt_nn(len2(a)); return l > 0 ? a * vrcp(l) : a; }

// CQuat::RotateVec3D (Vec3D.h:293-299) as a matrix: q f q* = M f for a unit quaternion (|q| = 1 to rounding).  A voxel's eight
// mesh corners go through the same rotation, so it is expanded once per voxel (§ Numerics).
struct RotFwd {
    double m[9];
    __device__ __forceinline__ RotFwd() {}
    __device__ __forceinline__ explicit RotFwd(dq q)
    {
        const double x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
        const double xx = q.x * x2, yy = q.y * y2, zz = q.z * z2, xy = q.x * y2, xz = q.x * z2, yz = q.y * z2, wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
        m[0] = 1 - (yy + zz); m[1] = xy - wz; m[2] = xz + wy;
        m[3] = xy + wz; m[4] = 1 - (xx + zz); m[5] = yz - wx;
        m[6] = xz - wy; m[7] = yz + wx; m[8] = 1 - (xx + yy);
    }
    __device__ __forceinline__ d3 operator()(d3 f) const
    { return mk3(m[0] * f.x + m[1] * f.y + m[2] * f.z, m[3] * f.x + m[4] * f.y + m[5] * f.z, m[6] * f.x + m[7] * f.y + m[8] * f.z); }
};


// drag of one facet (A, Bv, Cv) on its voxel moving with `speed` (sdir = speed normalised)
__device__ __forceinline__ d3 facet_drag_force(d3 speed, d3 sdir, d3 A, d3 Bv, d3 Cv, double drag_coef)
{
    const d3 AB = Bv - A, AC = Cv - A;
    const d3 cr = cross3(AB, AC);
    const double cl = vsqrt_nn(len2(cr));
    const double area = cl * 0.5;
    const d3 n = cl > 0 ? cr * vrcp(cl) : cr;           // CalcFaceNormals; the reference normalises the stored normal
                                                        // twice more before using it: identity to an ulp, skipped
    // LW/VX_Sim.cpp:1556-1559 tests (float)acos(c) < PI/2 with c = v^ . n^.  The largest float below PI/2 is
    // 1.57079625 and acos(c) rounds to it or below iff acos(c) <= 1.570796310901641845703125 (the midpoint
    // to the next float, a tie going to the even mantissa below), i.e. iff c >= cos(midpoint); c > 1
    // (two parallel unit vectors, rounding) makes the reference's acos a NaN and the facet drag-free.
    const double c = dot3(sdir, n);
    d3 contrib = mk3(0, 0, 0);
    if (c >= 1.5893254773528196e-08 && c <= 1.0) {
        const d3 proj = n * dot3(speed, n);             // ProjectOnTo
        // proj^ * (-k * area * |proj|^2) = proj * (-k * area * |proj|)
        contrib = proj * (-drag_coef * area * vsqrt_nn(len2(proj)));
    }
    return contrib;
}

struct VoxState { d3 pos, lm, am; dq ang; double scale; };

// CalcContactForce (VXS_BondCollision.cpp:41-59) of one listed partner at (qx, qy, qz), scale qs, on the voxel at `pos`: Force2 =
// unit(p2 - p1) * a1 * overlap on Vox2 (the later surface voxel), -Force2 on Vox1.  Seen from this voxel that is
// -unit(partner - me) * a1 * overlap in both roles, bit for bit (negation is exact, the sums commute).
// The function is inlined into every stepping kernel (resident: LDS rows and rows in memory; tiled; streaming); its roundings are
// pinned -- every product and sum rounded on its own, like the reference's x86-64 build, which has no fused multiply-add -- so
// that all of these sites produce the same bits: with contraction left to the compiler two sites of one kernel came out differently
// (measured: scripts/dev_gpu_diag.py contactcheck).  contact_in_reach is the reject test alone (resident kernel, pass 1).
#pragma clang fp contract(off)
__device__ __forceinline__ bool contact_in_reach(double dx, double dy, double dz, double nom)
{
    const double d2 = (dx * dx + dy * dy) + dz * dz;
    return d2 < nom * nom;
}
__device__ __forceinline__ d3 contact_force_add(d3 F, d3 pos, double scale, double qx, double qy, double qz, double qs, double a1)
{
    const double dx = qx - pos.x, dy = qy - pos.y, dz = qz - pos.z;
    const double nom = (qs + scale) * 0.75;
    const double d2 = (dx * dx + dy * dy) + dz * dz;
    if (d2 < nom * nom) {                                  // cheap reject: most listed partners are out of reach
        const double l = vsqrt_nn(d2);
        const double reld = nom - l;
        if (reld > 0) {
            const double il = vrcp(l);
            F.x = F.x - ((dx * il) * a1) * reld;
            F.y = F.y - ((dy * il) * a1) * reld;
            F.z = F.z - ((dz * il) * a1) * reld;
        }
    }
    return F;
}
#pragma clang fp contract(fast)

// position + scale of another voxel of the same robot, for the contact forces
struct FetchGlobal {       // streaming path: previous-step buffer in HBM
    const DBatch& B; int cur;
    __device__ __forceinline__ void operator()(int slot, double& x, double& y, double& z, double& s) const
    { x = POS(cur, 0, slot); y = POS(cur, 1, slot); z = POS(cur, 2, slot); s = SCALE(cur, slot); }
};
template <int BLOCK>
struct FetchLds {          // fused path: the workgroup's pose tile
    const double* ps; int base;
    __device__ __forceinline__ void operator()(int slot, double& x, double& y, double& z, double& s) const
    { const int l = slot - base; x = ps[l]; y = ps[BLOCK + l]; z = ps[2 * BLOCK + l]; s = ps[3 * BLOCK + l]; }
};

// per-voxel development parameters (float members of CVXS_Voxel, VXS_Voxel.h:92-111), only loaded for RF_DEV robots
struct DevParams { float init_size, final_size, start_growth, growth_time, phase, final_phase, final_tad; };
__device__ __forceinline__ DevParams load_dev(const DBatch& B, int v)
{
    DevParams d;
    const unsigned nv = B.nv;
    d.init_size = B.dev[v]; d.final_size = B.dev[nv + v]; d.start_growth = B.dev[2 * nv + v]; d.growth_time = B.dev[3 * nv + v];
    d.phase = B.dev[4 * nv + v]; d.final_phase = B.dev[5 * nv + v]; d.final_tad = B.dev[6 * nv + v];
    return d;
}

// Everything of EulerStep after the internal-bond sums, in two independent halves (a voxel's translation and its rotation + size
// never read each other's results within a step): voxel_update_lin -- collision bonds, gravity, drag, floor, linear integration --
// and voxel_update_ang -- angular integration, quaternion update, actuation -> new scale.  voxel_update runs one after the other on
// one lane (resident, tiled and streaming kernels); the wide kernel (kernels_wide.hpp) gives the halves of a small robot's voxels
// to different wavefronts.
// F arrives holding slow damping + internal bond forces.  `scale`: the voxel's size at the start of the step.  Returns |new velocity|^2.
template <class Fetch>
__device__ __forceinline__ double voxel_update_lin(const DBatch& B, const DRobot& R, const DVoxClass& C, int v, const Fetch& fetch,
                                                   d3 F, d3 vel, d3& pos, d3& lm, double scale, int row, int ccnt, bool fluid, d3 drag)
{
    // (v: global voxel slot = what the contact rows and `fetch` speak)
    const int flags = R.flags;
    const double dt = R.dt;
    if (ccnt > 0) {
        // collision bonds in creation order (VXS_Voxel.cpp:519-530, VXS_BondCollision.cpp:41-59); partner slots and the
        // pair stiffness a1 were stored by rebuild_rows.  Loads are issued four partners at a time.
        for (int k0 = 0; k0 < ccnt; k0 += 4) {
            int o[4]; double a1[4], qx[4], qy[4], qz[4], qs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = k0 + j < ccnt;
                const size_t at = col_at(R, k0 + j, row);                // partner-major: coalesced across the wave
                o[j] = on ? B.col_partner[at] : -1;
                a1[j] = on ? B.col_a1[at] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) fetch(o[j] < 0 ? v : o[j], qx[j], qy[j], qz[j], qs[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (o[j] < 0) continue;
                F = contact_force_add(F, pos, scale, qx[j], qy[j], qz[j], qs[j], a1[j]);
            }
        }
    }
    if ((flags & RF_GRAV) && !fluid) F.z += C.mass * R.grav_acc;
    if (fluid) F = F + drag;                  // LW/VXS_Voxel.cpp:372-373

    if ((flags & RF_FLOOR) && !fluid) {       // CalcFloorEffect, VXS_Voxel.cpp:708-758
        const double pen = 0.5 * scale - pos.z;
        bool static_fric = false;
        if (pen > 0) {
            const double normal = C.k_floor * pen;
            const double fz = normal - R.col_z * C.c_lin * vel.z;
            const double surf_vel = vsqrt_nn(vel.x * vel.x + vel.y * vel.y);
            const double surf_force = vsqrt_nn(F.x * F.x + F.y * F.y);
            const double fric = C.u_dynamic * normal;
            double fx = 0, fy = 0;
            bool stopped = (vel.x == 0 && vel.y == 0);
            if (flags & RF_STICKY) { lm.x = 0; lm.y = 0; static_fric = true; stopped = true; }
            if (stopped) {
                if (surf_force < C.u_static * normal) static_fric = true;
            } else if (fric * dt < C.mass * surf_vel) {
                // -(cos, sin)(atan2(vy, vx)) * fric == -(vx, vy)/|v| * fric
                const double inv = vdiv(fric, surf_vel);
                fx = -vel.x * inv; fy = -vel.y * inv;
            } else { static_fric = true; lm.x = 0; lm.y = 0; }
            F.x += fx; F.y += fy; F.z += fz;
        }
        if (static_fric) { F.x = 0; F.y = 0; }
    }

    // EulerStep, VXS_Voxel.cpp:183-189
    lm = lm + F * dt;
    pos = pos + lm * (dt * C.mass_inv);
    return len2(lm * C.mass_inv);
}

// M arrives holding minus the internal bond moments.  `scale` in: the size at the start of the step, out: the new one.
__device__ __forceinline__ void voxel_update_ang(const DBatch& B, const DRobot& R, const DVoxClass& C, int v, double t, double act_sin,
                                                 double act_cos, double prenatal_c, d3 M, d3& am, dq& q, double& scale,
                                                 double ph_sin, double ph_cos, float amp_damp)
{
    const int flags = R.flags;
    const double dt = R.dt;
    // EulerStep, VXS_Voxel.cpp:190-222
    am = am + M * dt;
    const double amf = 1 - 10 * R.slow_z * C.inertia_inv * C.c_ang * dt;
    am = am * amf;
    d3 w = am * C.inertia_inv;
    dq spin = qmul(mkq(0, w.x * 0.5, w.y * 0.5, w.z * 0.5), q);
    dq ang = mkq(q.w + spin.w * dt, q.x + spin.x * dt, q.y + spin.y * dt, q.z + spin.z * dt);
    {
        // NormalizeFast with the reference's own two roundings (sqrt, then 1 / l): the `w >= 1 -> identity` snap below
        // discards rotations smaller than ~1.5e-8 rad, and whether w lands on 1.0 depends on the last bit of 1 / l
        // (a single-rounding rsqrt here moved probe6 by 5e-9 voxel within ten steps)
        const double l = vsqrt_nn(ang.x * ang.x + ang.y * ang.y + ang.z * ang.z + ang.w * ang.w);
        if (l != 0) { const double li = vrcp(l); ang.w *= li; ang.x *= li; ang.y *= li; ang.z *= li; }
        if (ang.w >= 1.0) ang = mkq(1.0, 0, 0, 0);
    }
    q = ang;

    // thermal actuation -> new scale (VXS_Voxel.cpp:224-340 without development; LW/VXS_Voxel.cpp:211-235)
    // sin(2 pi' (t/T + phase)) of the reference = sin(a + b) with a = 2 pi' t/T, the same for every voxel (sincos once
    // per robot and step, actuation_sincos), and b = 2 pi' phase, fixed per voxel (DBatch::act_sb / act_cb)
    double new_scale;
    const double act = act_sin * ph_cos + act_cos * ph_sin;
    if (!(flags & RF_LW) && (flags & RF_DEV)) {
        // _voxcad with development layers (VXS_Voxel.cpp:236-340; the reference mixes float members and double locals, kept)
        const DevParams D = load_dev(B, v);
        const double nom = C.nom_size;
        const double prenatal = prenatal_c * (((double)D.init_size / nom) - 1);
        double dev_tf = 0, dphase = 0, dtad = 0, frozen = 0, freeze_init = 1;
        if (R.midlife_freeze_time > 0) {                                 // :247-264
            const double middle = 0.5 * (R.stop_value - R.init_cm_time);
            const double fs = middle - 0.5 * R.midlife_freeze_time, fe = middle + 0.5 * R.midlife_freeze_time;
            if (t > fs && t < fe) { frozen = t - fs; if (t < fs + R.init_cm_time) freeze_init = 0; }
            if (t > fe) frozen = R.midlife_freeze_time;
        }
        if (t >= (double)D.start_growth && D.growth_time > 0) {          // postnatal linear development :267-296
            const float sg = D.start_growth + D.growth_time;             // float sum, as in the reference
            double eff = (t <= (double)sg + R.midlife_freeze_time) ? t : (double)sg + R.midlife_freeze_time;
            eff = eff - frozen;
            const double k = (eff - (double)D.start_growth) / (double)D.growth_time;
            if (flags & RF_DEV_FSIZE) { const float ratio = D.final_size / D.init_size; dev_tf = k * ((double)ratio - 1.0); }
            if (flags & RF_DEV_FPHASE) { const float d = D.final_phase - D.phase; dphase = k * (double)d; }
            if (flags & RF_DEV_FTAD) { const float d = D.final_tad - amp_damp; dtad = k * (double)d; }
        }
        double ctrl = 0;
        if ((flags & RF_TEMP) && t >= R.init_cm_time)
            ctrl = ((double)amp_damp + dtad) * ((double)R.temp_amplitude * sin((double)(2 * 3.1415926f) * (t / (double)R.temp_period + ((double)D.phase + dphase)))) * C.cte * freeze_init;
        if (flags & RF_DEV_SIZE) {                                       // actuation limited by the current size :316-328
            const double cur_size = (1 + prenatal) * (1 + dev_tf) * nom;
            const double sig = ((cur_size / nom - 1) / R.growth_amplitude + 1) * 0.5;
            ctrl = ctrl * (sig > 0.5 ? 0.5 : sig) * 2;
        }
        new_scale = ctrl * nom + (1 + prenatal) * (1 + dev_tf) * nom;
        const double max_scale = (1 + R.growth_amplitude) * nom, min_scale = R.min_temp_fact * nom;
        if (new_scale < scale && new_scale < min_scale) new_scale = scale;
        if (new_scale > scale && new_scale > max_scale) new_scale = scale;
    } else if (!(flags & RF_LW)) {
        const double prenatal = prenatal_c * C.prenatal_k;       // prenatal_c * ((float)nom / nom - 1), the quotient formed at import (same rounding)
        double ctrl = 0;
        if ((flags & RF_TEMP) && t >= R.init_cm_time)
            ctrl = (double)amp_damp * ((double)R.temp_amplitude * act) * C.cte;
        new_scale = ctrl * C.nom_size + (1 + prenatal) * C.nom_size;
        const double max_scale = (1 + R.growth_amplitude) * C.nom_size, min_scale = R.min_temp_fact * C.nom_size;
        if (new_scale < scale && new_scale < min_scale) new_scale = scale;
        if (new_scale > scale && new_scale > max_scale) new_scale = scale;
    } else {
        double tf = 1.0;
        if ((flags & RF_TEMP) && t >= R.init_cm_time)
            tf = 1 + ((double)R.temp_amplitude * act) * C.cte;
        if (tf < 0.1) tf = 0.1;
        new_scale = tf * C.nom_size;
    }
    scale = new_scale;
}

template <class Fetch>
__device__ __forceinline__ double voxel_update(const DBatch& B, const DRobot& R, const DVoxClass& C, int v, const Fetch& fetch,
                                               double t, double act_sin, double act_cos, double prenatal_c, d3 F, d3 M, d3 vel,
                                               VoxState& S, int row, int ccnt, bool fluid, d3 drag, double ph_sin, double ph_cos, float amp_damp)
{
    const double vel2 = voxel_update_lin(B, R, C, v, fetch, F, vel, S.pos, S.lm, S.scale, row, ccnt, fluid, drag);
    voxel_update_ang(B, R, C, v, t, act_sin, act_cos, prenatal_c, M, S.am, S.ang, S.scale, ph_sin, ph_cos, amp_damp);
    return vel2;
}

// uniform per-step factors of the actuation, evaluated once per robot and step instead of once per voxel
__device__ __forceinline__ void actuation_sincos(const DRobot& R, double t, double& s, double& c)
{
    s = 0; c = 0;
    if ((R.flags & RF_TEMP) && t >= R.init_cm_time) sincos((double)(2 * 3.1415926f) * (t / (double)R.temp_period), &s, &c);
}
__device__ __forceinline__ double actuation_prenatal_c(const DRobot& R, double t) { return (t >= 0.5 * R.init_cm_time) ? 1.0 : 2 * t / R.init_cm_time; }

// ------------------------------------------------------------------------------------- per-robot step control
struct StepCtl { int go, latch, eol, rebuild, trace, trace_index; };

// Executed by ONE thread per robot before every step (and once after the last): completes the previous step's
// accounting, evaluates the stop condition and decides what this step needs.  Two parts: step_control_begin needs
// nothing of the previous step's voxel phase; step_control_horizon consumes its MaxVoxVel.
__device__ __forceinline__ StepCtl step_control_begin(const DRobot& R, DRobotState& rs, long long step_cap, int begin_new_step)
{
    StepCtl c; c.go = c.latch = c.eol = c.rebuild = c.trace = c.trace_index = 0;
    if (rs.status != 0) return c;
    if (rs.active) {                         // finish the step the previous round computed
        if (rs.diverged) rs.status = 2;
        else {
            rs.cur_time += R.dt; rs.steps += 1; rs.dt_prev = R.dt;
            // UpdateStats of that step (VX_Sim.cpp:1537-1547): a point of the centre-of-mass trace is due; the pass that latches
            // IniCM computes it from the poses the step left (also when the robot stops now: c.go stays 0, c.trace is set)
            if (R.trace_dt > 0 && rs.cur_time > R.init_cm_time && (rs.ntrace == 0 || rs.last_trace_time + R.trace_dt <= rs.cur_time)) {
                if (rs.ntrace < R.trace_cap) { c.trace = 1; c.trace_index = rs.ntrace; }
                rs.ntrace += 1; rs.last_trace_time = rs.cur_time;
            }
        }
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
    rs.active = 1;
    return c;
}
// (dt_prev: the step length of the step before the one being decided -- rs.dt_prev, unless the caller has already let
// step_control_begin of the NEXT step overwrite it)
__device__ __forceinline__ void step_control_horizon(const DRobot& R, DRobotState& rs, StepCtl& c, double dt_prev)
{
    if (c.go && (R.flags & RF_SELF_COL)) {                                       // UpdateCollisions :1729-1755
        // (lean sqrt / divide: bit-identical for the finite non-negative operands here, a third of the instructions -- this runs
        // on one thread while its workgroup waits)
        const double mv = vsqrt_nn(__longlong_as_double((long long)rs.maxvel2_bits));
        rs.max_disp += fabs(vdiv(mv * dt_prev, R.lat));
        rs.maxvel2_bits = 0ull;
        if (!(R.flags & RF_HORIZON_COL) || rs.max_disp > (R.col_horizon - 1.0) / 2) { c.rebuild = 1; rs.max_disp = 0.0; rs.rebuilds += 1; rs.col_tiled = 0; rs.reb_step = rs.steps; }
    }
    rs.rebuild_now = c.rebuild;
}
__device__ __forceinline__ StepCtl step_control(const DRobot& R, DRobotState& rs, long long step_cap, int begin_new_step)
{
    StepCtl c = step_control_begin(R, rs, step_cap, begin_new_step);
    step_control_horizon(R, rs, c, rs.dt_prev);
    return c;
}

// ================================================================================================ streaming path
// where the whole-robot passes (IniCM latch, broad-phase) read position + scale of an arbitrary voxel of the robot
struct PoseFromState {     // streaming path: the state planes, buffer `cur`
    const DBatch& B; int cur;
    __device__ __forceinline__ void operator()(int slot, double& x, double& y, double& z, double& s) const
    { x = POS(cur, 0, slot); y = POS(cur, 1, slot); z = POS(cur, 2, slot); s = SCALE(cur, slot); }
};
// IniCM latch (= SS.CurCM of the previous step: mass-weighted SEQUENTIAL sum in voxel order, GetCM VX_Sim.cpp:2415-2430)
// and EndOfLifetimePosteriorY (getPosteriorY :2640-2656).  Whole workgroup; `sh` holds 5*CH doubles of LDS scratch.
template <class Pose>
__device__ __forceinline__ void latch_cm(const DBatch& B, const DRobot& R, DRobotState& rs, const Pose& pose, bool latch, bool eol, double* sh, int CH,
                                         bool trace = false, int trace_index = 0)
{
    // The sums keep the reference's order (one thread adds voxel after voxel), but only the additions are serial: the products
    // x * m are formed by the staging threads (the same single rounding), and the y / lat of getPosteriorY is taken once, of the
    // smallest y (v -> v / lat is monotone for lat > 0, so the minimum of the rounded quotients is the rounded quotient of the minimum)
    const int tid = threadIdx.x, T = blockDim.x, base = R.vox_begin;
    double sx = 0, sy = 0, sz = 0, sm = 0, miny = 1.0e300;
    for (int c0 = 0; c0 < R.nvox; c0 += CH) {
        for (int k = tid; k < CH && c0 + k < R.nvox; k += T) {
            const int g = base + c0 + k;
            const DVoxClass& C = B.vclass_tab[R.vtab_begin + B.vclass[g]];
            double x, y, z, unused_scale;
            pose(g, x, y, z, unused_scale);
            sh[k] = __dmul_rn(x, C.mass); sh[CH + k] = __dmul_rn(y, C.mass); sh[2 * CH + k] = __dmul_rn(z, C.mass);
            sh[3 * CH + k] = C.mass;
            sh[4 * CH + k] = (C.mat == 5) ? 1.0e300 : y;        // material 5 is excluded from PosteriorY
        }
        __syncthreads();
        if (tid == 0) {
            const int n = min(CH, R.nvox - c0);
            for (int k = 0; k < n; ++k) {
                sx = __dadd_rn(sx, sh[k]); sy = __dadd_rn(sy, sh[CH + k]); sz = __dadd_rn(sz, sh[2 * CH + k]); sm += sh[3 * CH + k];
                miny = sh[4 * CH + k] < miny ? sh[4 * CH + k] : miny;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (latch) { const double inv = 1.0 / sm; rs.ini_cm[0] = inv * sx; rs.ini_cm[1] = inv * sy; rs.ini_cm[2] = inv * sz; rs.cm_init = 1; }
        if (eol) { const double q = miny / R.lat; rs.eol_post_y = q < 100000.0 ? q : 100000.0; }
        if (trace) {                         // SS.CMTraceTime / SS.CMTrace: (CurTime, GetCM()) of the step just finished
            const double inv = 1.0 / sm;
            double* e = B.trace + (size_t)(R.trace_begin + trace_index) * 4;
            e[0] = rs.cur_time; e[1] = inv * sx; e[2] = inv * sy; e[3] = inv * sz;
        }
    }
}

// The reductions behind the result file, from the state the stepping kernels left in HBM: one workgroup per robot.  The centre of
// mass keeps the reference's summation order (thread 0 adds voxel after voxel; the products come from all threads); the extrema
// and counts are order-free.  Every expression that the host used to evaluate is written with explicit roundings (no contraction),
// so the numbers are those of the host path bit for bit (option host_results = 1 keeps that path for cross-checks).
static __global__ __launch_bounds__(256) void k_results(DBatch B, DResult* __restrict__ out)
{
#pragma clang fp contract(off)      // every product and sum below is rounded on its own, like on the host (HIP's *_rn arithmetic helpers would not
                                    // do: they are plain operators compiled where they are DEFINED, with contraction on)
    const int r = blockIdx.x, tid = threadIdx.x;
    const DRobot& R = B.robot[r];
    const DRobotState& rs = B.rstate[r];
    __shared__ double sh[4 * 256];
    __shared__ double red[4][4];
    __shared__ int cnt[2][4];
    const int cur = rs.steps & 1, base = R.vox_begin;
    double sx = 0, sy = 0, sz = 0, sm = 0;
    double d2max = 0.0, d2min = 1.0e300, ymax = -1.0e300, ymin = 1.0e300;
    int touching = 0, feet = 0;
    const double ix = rs.ini_cm[0], iy = rs.ini_cm[1];
    for (int c0 = 0; c0 < R.nvox; c0 += 256) {
        const int k = c0 + tid;
        if (k < R.nvox) {
            const int g = base + k;
            const DVoxClass& C = B.vclass_tab[R.vtab_begin + B.vclass[g]];
            const double x = POS(cur, 0, g), y = POS(cur, 1, g), z = POS(cur, 2, g), s = SCALE(cur, g);
            sh[tid] = x * C.mass; sh[256 + tid] = y * C.mass; sh[512 + tid] = z * C.mass; sh[768 + tid] = C.mass;
            const double dx = x - ix, dy = y - iy;
            const double dxx = dx * dx, dyy = dy * dy;
            const double d2 = dxx + dyy;
            d2max = d2 > d2max ? d2 : d2max; d2min = d2 < d2min ? d2 : d2min;
            if (C.mat != 5) { ymax = y > ymax ? y : ymax; ymin = y < ymin ? y : ymin; }
            const double half = 0.5 * s;
            if (half - z > 0) { ++touching; if (C.mat == 6) ++feet; }
        }
        __syncthreads();
        if (tid == 0) {
            const int n = min(256, R.nvox - c0);
            for (int j = 0; j < n; ++j) { sx = sx + sh[j]; sy = sy + sh[256 + j]; sz = sz + sh[512 + j]; sm = sm + sh[768 + j]; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double a = __shfl_xor(d2max, off), b = __shfl_xor(d2min, off), c = __shfl_xor(ymax, off), d = __shfl_xor(ymin, off);
        d2max = a > d2max ? a : d2max; d2min = b < d2min ? b : d2min; ymax = c > ymax ? c : ymax; ymin = d < ymin ? d : ymin;
        touching += __shfl_xor(touching, off); feet += __shfl_xor(feet, off);
    }
    if ((tid & 63) == 0) { const int w = tid >> 6; red[0][w] = d2max; red[1][w] = d2min; red[2][w] = ymax; red[3][w] = ymin; cnt[0][w] = touching; cnt[1][w] = feet; }
    __syncthreads();
    if (tid == 0) {
        DResult o;
        const double inv = R.nvox > 0 ? 1.0 / sm : 0.0;
        o.cm[0] = inv * sx; o.cm[1] = inv * sy; o.cm[2] = inv * sz;
        o.d2max = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]));
        o.d2min = fmin(fmin(red[1][0], red[1][1]), fmin(red[1][2], red[1][3]));
        o.ymax = fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3]));
        o.ymin = fmin(fmin(red[3][0], red[3][1]), fmin(red[3][2], red[3][3]));
        o.touching = cnt[0][0] + cnt[0][1] + cnt[0][2] + cnt[0][3];
        o.feet = cnt[1][0] + cnt[1][1] + cnt[1][2] + cnt[1][3];
        out[r] = o;
    }
}

// stiffness of the collision bond between two voxel classes, first = the earlier voxel (CVX_Bond::LinkVoxels +
// UpdateConstants for the pair, VX_Bond.cpp:65-173: a1 = E*A/L of a cubic bond)
__device__ __forceinline__ double contact_a1(const DVoxClass& C1, const DVoxClass& C2)
{
    const double E = (C1.E * C2.E / (C1.E + C2.E)) * 2;
    const double L = (C1.nom_size + C2.nom_size) * 0.5;
    return E * (L * L) / L;
}

// CalcL1Bonds (VX_Sim.cpp:2357-2413).  The calling workgroup builds the partner rows of the surface voxels its
// threads own: every surface voxel tests all others (staged through LDS in chunks of
// CH) and keeps, in ascending partner order = creation order of its collision bonds in the reference, those that
// pass the distance filter, are more than `hops` bonds away and lie within CollisionHorizon scaled voxel sizes.
// `sh`: 4*CH doubles + CH ints of LDS.
// Thread `tid` owns surface ordinal i (`mine` = it has one).
template <class Pose>
__device__ __forceinline__ void rebuild_rows(const DBatch& B, const DRobot& R, DRobotState& rs, const Pose& pose, int i, bool mine, double* sh, int CH)
{
    const int tid = threadIdx.x, T = blockDim.x;
    int* shv = (int*)(sh + 4 * CH);
    int vi = 0, cnt = 0;
    const unsigned long long* row = B.excl + R.excl_begin + (long long)(mine ? i : 0) * R.excl_wpr;
    d3 pi = mk3(0, 0, 0); double si = 0;
    if (mine) {
        vi = B.surf[R.surf_begin + i];
        pose(vi, pi.x, pi.y, pi.z, si);
    }
    const double H = R.col_horizon;
    for (int c0 = 0; c0 < R.nsurf; c0 += CH) {
        const int n = min(CH, R.nsurf - c0);
        for (int k = tid; k < n; k += T) {
            const int vj = B.surf[R.surf_begin + c0 + k];
            pose(vj, sh[k], sh[CH + k], sh[2 * CH + k], sh[3 * CH + k]); shv[k] = vj;
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
                    if (cnt < R.col_cap) {
                        const int vj = shv[k];
                        const DVoxClass& Ci = B.vclass_tab[R.vtab_begin + B.vclass[vi]];
                        const DVoxClass& Cj = B.vclass_tab[R.vtab_begin + B.vclass[vj]];
                        const size_t at = col_at(R, cnt, R.surf_begin + i);
                        B.col_partner[at] = vj;
                        B.col_a1[at] = (j > i) ? contact_a1(Ci, Cj) : contact_a1(Cj, Ci);
                    }
                    ++cnt;
                }
            }
        }
        __syncthreads();
    }
    if (mine) {
        if (cnt > R.col_cap) { cnt = R.col_cap; atomicOr(&rs.col_overflow, 1); }
        B.col_cnt[R.surf_begin + i] = cnt;
    }
}

static __global__ __launch_bounds__(256) void k_step_begin(DBatch B, long long step_cap, int begin_new_step)
{
    const int r = blockIdx.x;
    if (!B.streamed[r]) return;             // (stepped by the resident kernel)
    const DRobot& R = B.robot[r];
    DRobotState& rs = B.rstate[r];
    __shared__ double sh[5 * 256];
    __shared__ int s_latch, s_eol, s_trace, s_tidx;
    if (threadIdx.x == 0) {
        StepCtl c = step_control(R, rs, step_cap, begin_new_step);
        s_latch = c.latch; s_eol = c.eol; s_trace = c.trace; s_tidx = c.trace_index;
        if (c.go) actuation_sincos(R, rs.cur_time, rs.act_sin, rs.act_cos);
    }
    __syncthreads();
    if (s_latch || s_eol || s_trace) latch_cm(B, R, rs, PoseFromState{B, rs.steps & 1}, s_latch != 0, s_eol != 0, sh, 256, s_trace != 0, s_tidx);
}

template <int A>
__device__ __forceinline__ void stream_bond(const DBatch& B, const DRobot& R, const DRobotState& rs, int r, int slot, int v1, int bc)
{
    const int v2 = B.nbr[(2 * A) * B.nv + v1];
    const int cur = rs.steps & 1;
    d3 p1 = mk3(POS(cur, 0, v1), POS(cur, 1, v1), POS(cur, 2, v1));
    d3 p2 = mk3(POS(cur, 0, v2), POS(cur, 1, v2), POS(cur, 2, v2));
    dq q1 = mkq(QUAT(0, v1), QUAT(1, v1), QUAT(2, v1), QUAT(3, v1));
    dq q2 = mkq(QUAT(0, v2), QUAT(1, v2), QUAT(2, v2), QUAT(3, v2));
    const double sc1 = SCALE(cur, v1), sc2 = SCALE(cur, v2);
    BondHist H = load_bond_hist(B, slot);
    const unsigned old_flags = H.flags;
    BondOut o = bond_compute<A>(B, B.bclass_tab[R.btab_begin + bc], H, p1, q1, sc1, p2, q2, sc2, rs.dt_prev != 0);
    store_bond_hist(B, slot, H, old_flags);
    if (o.diverged) atomicOr(&B.rstate[r].diverged, 1);
    if (R.nmv > 0) {      // land_water: CurStrainV1 / CurStrainV2 (SetStrainDir), inputs of the surface mesh in the next step
        B.strain[(unsigned)A * B.nv + v1] = o.strain1;
        B.strain[(unsigned)(3 + A) * B.nv + v2] = o.strain2;
    }
    BOUT(0, slot) = o.f1.x; BOUT(1, slot) = o.f1.y; BOUT(2, slot) = o.f1.z;
    BOUT(3, slot) = o.m1.x; BOUT(4, slot) = o.m1.y; BOUT(5, slot) = o.m1.z;
    BOUT(6, slot) = o.f2.x; BOUT(7, slot) = o.f2.y; BOUT(8, slot) = o.f2.z;
    BOUT(9, slot) = o.m2.x; BOUT(10, slot) = o.m2.y; BOUT(11, slot) = o.m2.z;
}

// Fluid drag of the streaming path (robots in a fluid that do not fit the resident kernel): the surface mesh of the step,
// one thread per vertex, then one thread per facet; k_voxels adds up each voxel's facets.  Launched between k_step_begin
// and k_bonds: the strains read here are those of the previous step, the poses and momenta those at the start of this one.
static __global__ __launch_bounds__(256) void k_mesh_vertices(DBatch B)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B.n_mv) return;
    const int r = B.vert_robot[i];
    const DRobot& R = B.robot[r];
    const DRobotState& rs = B.rstate[r];
    if (!B.streamed[r] || !(R.flags & RF_FLUID) || !rs.active) return;
    const int cur = rs.steps & 1;
    const unsigned tm = B.total_mv, nv = B.nv;
    const double nom = R.lat;
    d3 part = mk3(0, 0, 0);
    int count = 0;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {              // corner-code order, as in fused_drag
        const int g = B.vert_vox[(unsigned)corner * tm + i];
        if (g < 0) continue;
        const double hx = (1 + B.strain[((corner & 4) ? 0u : 3u) * nv + g]) * nom * 0.5;     // CornerPosCur / CornerNegCur
        const double hy = (1 + B.strain[((corner & 2) ? 1u : 4u) * nv + g]) * nom * 0.5;
        const double hz = (1 + B.strain[((corner & 1) ? 2u : 5u) * nv + g]) * nom * 0.5;
        const RotFwd M(mkq(QUAT(0, g), QUAT(1, g), QUAT(2, g), QUAT(3, g)));
        part = part + (mk3(POS(cur, 0, g), POS(cur, 1, g), POS(cur, 2, g)) + M(mk3((corner & 4) ? hx : -hx, (corner & 2) ? hy : -hy, (corner & 1) ? hz : -hz)));
        ++count;
    }
    const double inv = vrcp((double)count);
    const d3 v0 = mk3(B.vert_v0[i], B.vert_v0[tm + i], B.vert_v0[2u * tm + i]);
    const d3 np = part * inv;
    const d3 now = v0 + (np - v0);                            // v + DrawOffset, as the reference stores it
    B.mesh_pos[i] = now.x; B.mesh_pos[tm + i] = now.y; B.mesh_pos[2u * tm + i] = now.z;
}

static __global__ __launch_bounds__(256) void k_facets(DBatch B)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= B.n_facet) return;
    const int r = B.facet_robot[f];
    const DRobot& R = B.robot[r];
    const DRobotState& rs = B.rstate[r];
    if (!B.streamed[r] || !(R.flags & RF_FLUID) || !rs.active) return;
    const unsigned tm = B.total_mv, tf = B.total_facet;
    const int u = R.vox_begin + B.facet_vox[f];
    const d3 speed = mk3(LINMOM(0, u), LINMOM(1, u), LINMOM(2, u)) * B.vclass_tab[R.vtab_begin + B.vclass[u]].mass_inv;
    const unsigned ia = R.vert_begin + B.facet_vert[f], ib = R.vert_begin + B.facet_vert[tf + f], ic = R.vert_begin + B.facet_vert[2u * tf + f];
    const d3 contrib = facet_drag_force(speed, normalized3(speed), mk3(B.mesh_pos[ia], B.mesh_pos[tm + ia], B.mesh_pos[2u * tm + ia]),
                                        mk3(B.mesh_pos[ib], B.mesh_pos[tm + ib], B.mesh_pos[2u * tm + ib]),
                                        mk3(B.mesh_pos[ic], B.mesh_pos[tm + ic], B.mesh_pos[2u * tm + ic]), R.drag_coef);
    B.fdrag[f] = contrib.x; B.fdrag[tf + f] = contrib.y; B.fdrag[2u * tf + f] = contrib.z;
}

// blocks [0, bond_blocks): one thread per bond slot; blocks beyond: collision-list rebuilds (reb_robot/reb_i0 tables),
// which overlap with the bond work of the other robots
static __global__ __launch_bounds__(256) void k_bonds(DBatch B, int bond_blocks, const int* __restrict__ reb_robot, const int* __restrict__ reb_i0)
{
    if ((int)blockIdx.x >= bond_blocks) {
        __shared__ double sh[4 * 512 + 256];
        const int k = blockIdx.x - bond_blocks;
        const int r = reb_robot[k];
        DRobotState& rs = B.rstate[r];
        if (!B.streamed[r] || !rs.active || !rs.rebuild_now) return;
        const int i = reb_i0[k] + (int)threadIdx.x;
        rebuild_rows(B, B.robot[r], rs, PoseFromState{B, rs.steps & 1}, i, i < B.robot[r].nsurf, sh, 512);
        return;
    }
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= 3 * B.nv) return;
    const int axis = tid / B.nv;            // wave-uniform: nv is a multiple of 64
    const int v1 = tid - axis * B.nv;
    const int r = robot_of(B, v1);
    if (r < 0 || !B.streamed[r]) return;
    const DRobotState& rs = B.rstate[r];
    if (!rs.active) return;
    const int bc = B.bclass[tid];
    if (bc < 0) return;
    const DRobot& R = B.robot[r];
    if (axis == 0) stream_bond<0>(B, R, rs, r, tid, v1, bc);
    else if (axis == 1) stream_bond<1>(B, R, rs, r, tid, v1, bc);
    else stream_bond<2>(B, R, rs, r, tid, v1, bc);
}

static __global__ __launch_bounds__(256) void k_voxels(DBatch B)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= B.nv) return;
    const int r = robot_of(B, v);
    if (r < 0 || !B.streamed[r]) return;
    DRobotState& rs = B.rstate[r];
    if (!rs.active || rs.diverged) return;    // Integrate() returns before the voxel loop when a bond diverged
    const DRobot& R = B.robot[r];
    const bool valid = (v - R.vox_begin) < R.nvox;   // padding slots stay in the wave for the reduction below
    double vel2 = 0;
    if (valid) {
        const DVoxClass& C = B.vclass_tab[R.vtab_begin + B.vclass[v]];
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
        const FetchGlobal fetch{B, cur};
        const bool fluid = (R.flags & RF_FLUID) != 0;
        d3 drag = mk3(0, 0, 0);
        if (fluid) {                          // my facets in the reference's order (k_facets ran earlier in this step)
            const unsigned tf = B.total_facet, f0 = R.facet_begin + B.facet_first[v];
            for (int j = 0; j < (int)B.facet_count[v]; ++j) drag = drag + mk3(B.fdrag[f0 + j], B.fdrag[tf + f0 + j], B.fdrag[2u * tf + f0 + j]);
        }
        vel2 = voxel_update(B, R, C, v, fetch, rs.cur_time, rs.act_sin, rs.act_cos, actuation_prenatal_c(R, rs.cur_time), F, M, vel, S, row, ccnt, fluid,
                             drag, B.act_sb[v], B.act_cb[v], B.amp_damp[v]);
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

#include "kernels_fused.hpp"
#ifdef VXH_PAIR      // measured 20-30 % slower than the kernels it replaces (DESIGN.md "Pair path"): kept for A/B in the developer library (make prof)
#include "kernels_pair.hpp"
#endif
#include "kernels_wide.hpp"
#include "kernels_tiled.hpp"
