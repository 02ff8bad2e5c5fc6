// SMPL linear blend skinning forward + backward on gfx950 (K7 of SURVEY.md §2.2).
//
// Replaces tuch/models/smpl.py:44-56, i.e. smplx 0.1.13 SMPL.forward -> lbs() (third-party,
// SURVEY.md §3.3) + the 9 extra regressed joints and the 49-joint re-map:
//   v_shaped = v_template + shapedirs.beta          J = J_regressor.v_shaped
//   R = rodrigues(pose)   v_posed = v_shaped + posedirs^T.(R[1:] - I)
//   (posed joints, A) = rigid chain(R, J)            verts = (sum_j W_vj A_j) [v_posed; 1]
//   joints = cat(posed joints, verts[picked 21], J_regressor_extra.verts)[joint_map]
//
// Kernels (f32 throughout; MFMA = v_mfma_f32_16x16x4_f32, exact f32 FMA chains):
//   pose_kernel        per body: Rodrigues, J = J_template + J_shapedirs.beta (the joint regressor
//                      folded through the shape basis once at model creation), kinematic chain,
//                      blend-feature row [R[1:]-I | beta | 1].
//   blend_kernel       MFMA GEMM  v_posed[B, 3V] = feat[B,220] x [posedirs; shapedirs^T; v_template]
//   skin_kernel        per (body, vertex): T = sum_j W_vj A_j (A wave-uniform, scalar loads), verts
//                      + per-workgroup partial sums of the extra-joint regression J_regressor_extra[9,V] x verts
//   assemble_kernel    joints[49] via joint_map
// Backward mirrors it: joints scatter, skinning adjoint (g_v_posed = T_R^T g_v and
// g_A = W^T (g_v (x) [v_posed;1]) on MFMA), blend adjoint (MFMA split-K GEMM against the same
// matrix), chain + Rodrigues adjoint per body.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct tuch_smpl_model {
    int V, N3;                 // vertices, 3V
    int N3p;                   // row stride of blend / v_posed / g_vposed: 3V rounded up to 64 floats (rows start on 256 bytes)
    float* blend;              // [kFeatRows][N3p]: posedirs (207) | shapedirs^T (10) | v_template | zero rows; zero padding columns
    float* J_template;         // [24*3]
    float* J_shapedirs;        // [24*3][10]
    float* weights;            // [V][24]
    float* weights_t;          // [24][V]: the same by joint -- a wavefront's 64 vertices read one joint's weights from two
                               // cache lines; by vertex (96 bytes apart) every load touched 64 lines
    int skin_nnz;              // largest number of non-zero skinning weights of a vertex; <= 4: the sparse tables below exist
    int32_t* skin_joint;       // [4][V]: the vertex's joints with a non-zero weight in ascending order (padding: joint 0) ...
    float* skin_weight;        // [4][V]: ... and their weights (padding: 0).  SMPL's own lbs_weights have at most 4 per vertex
    float* Jrx;                // [9][V]  J_regressor_extra
    int32_t* parents;          // [48]: parent of joint k (-1 for the root) | depth of joint k in the tree
    int max_depth;
    int32_t* extra_ids;        // [21]
    int32_t* joint_map;        // [49]
    int parents_host[24];
};

namespace {

constexpr int kJoints = 24;
constexpr int kBetas = 10;
constexpr int kPoseFeat = 207;
constexpr int kFeat = 220;            // 207 + 10 + 1, padded to a multiple of 4
constexpr int kFeatRows = 224;        // rows of the matrix in memory: the forward's four K quarters are 56 rows each
constexpr int kPicked = 21;
constexpr int kExtra = 9;
constexpr int kAllJoints = kJoints + kPicked + kExtra;   // 54
constexpr int kOutJoints = 49;
constexpr int kSkinBlock = 256;

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---------------------------------------------------------------------------------- 3x3 helpers
struct M3 { float m[9]; };
__device__ __forceinline__ M3 mul(const M3& a, const M3& b)
{
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return r;
}
__device__ __forceinline__ M3 mul_nt(const M3& a, const M3& b)   // a * b^T
{
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r.m[3 * i + j] = a.m[3 * i] * b.m[3 * j] + a.m[3 * i + 1] * b.m[3 * j + 1] + a.m[3 * i + 2] * b.m[3 * j + 2];
    return r;
}
__device__ __forceinline__ M3 mul_tn(const M3& a, const M3& b)   // a^T * b
{
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r.m[3 * i + j] = a.m[i] * b.m[j] + a.m[3 + i] * b.m[3 + j] + a.m[6 + i] * b.m[6 + j];
    return r;
}
__device__ __forceinline__ void mulv(const M3& a, const float* v, float* o)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = a.m[3 * i] * v[0] + a.m[3 * i + 1] * v[1] + a.m[3 * i + 2] * v[2];
}
__device__ __forceinline__ void mulv_t(const M3& a, const float* v, float* o)   // a^T v
{
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = a.m[i] * v[0] + a.m[3 + i] * v[1] + a.m[6 + i] * v[2];
}

// smplx batch_rodrigues: angle = |aa + 1e-8|, d = aa/angle, R = I + sin K + (1-cos) K^2
__device__ __forceinline__ M3 rodrigues(const float* aa)
{
    const float ex = aa[0] + 1e-8f, ey = aa[1] + 1e-8f, ez = aa[2] + 1e-8f;
    const float th = sqrtf(ex * ex + ey * ey + ez * ez);
    const float x = aa[0] / th, y = aa[1] / th, z = aa[2] / th;
    const float s = sinf(th), c1 = 1.0f - cosf(th);
    // K^2 = d d^T - |d|^2 I
    const float dd = x * x + y * y + z * z;
    M3 r;
    r.m[0] = 1.0f + c1 * (x * x - dd); r.m[1] = -s * z + c1 * x * y;      r.m[2] = s * y + c1 * x * z;
    r.m[3] = s * z + c1 * x * y;       r.m[4] = 1.0f + c1 * (y * y - dd); r.m[5] = -s * x + c1 * y * z;
    r.m[6] = -s * y + c1 * x * z;      r.m[7] = s * x + c1 * y * z;       r.m[8] = 1.0f + c1 * (z * z - dd);
    return r;
}

// adjoint of rodrigues(): gR -> g_aa
__device__ __forceinline__ void rodrigues_bwd(const float* aa, const M3& g, float* g_aa)
{
    const float ex = aa[0] + 1e-8f, ey = aa[1] + 1e-8f, ez = aa[2] + 1e-8f;
    const float th = sqrtf(ex * ex + ey * ey + ez * ez);
    const float d[3] = {aa[0] / th, aa[1] / th, aa[2] / th};
    const float s = sinf(th), c = cosf(th), c1 = 1.0f - c;
    const M3 K = {{0.f, -d[2], d[1], d[2], 0.f, -d[0], -d[1], d[0], 0.f}};
    const M3 K2 = mul(K, K);
    float g_th = 0.f;
#pragma unroll
    for (int e = 0; e < 9; ++e) g_th += g.m[e] * (c * K.m[e] + s * K2.m[e]);
    // gK = s G + (1-c) (G K^T + K^T G)
    const M3 gkt = mul_nt(g, K), ktg = mul_tn(K, g);
    M3 gK;
#pragma unroll
    for (int e = 0; e < 9; ++e) gK.m[e] = s * g.m[e] + c1 * (gkt.m[e] + ktg.m[e]);
    const float gd[3] = {gK.m[7] - gK.m[5], gK.m[2] - gK.m[6], gK.m[3] - gK.m[1]};
    g_th += -(gd[0] * aa[0] + gd[1] * aa[1] + gd[2] * aa[2]) / (th * th);
    g_aa[0] = gd[0] / th + g_th * ex / th;
    g_aa[1] = gd[1] / th + g_th * ey / th;
    g_aa[2] = gd[2] / th + g_th * ez / th;
}

// The pose as the callers hold it: the root rotation (global_orient) and the 23 body rotations (body_pose) are two
// tensors in the reference's API (models/smpl.py:44-47 concatenates them); joint t of body b is read from / its gradient
// written to the right one directly.  w = 3 (axis-angle) or 9 (rotation matrix) floats per joint; strides in floats.
struct PoseRef { const float* root; const float* body; int root_stride, body_stride; };
// body_add (or nullptr): a gradient the caller already holds for body_pose, added to what is written (no separate add launch)
struct PoseGrad { float* root; float* body; int root_stride, body_stride; const float* body_add; int body_add_stride; };
__device__ __forceinline__ const float* pose_joint(const PoseRef& p, int b, int t, int w)
{
    return t == 0 ? p.root + (size_t)b * p.root_stride : p.body + (size_t)b * p.body_stride + (size_t)(t - 1) * w;
}
__device__ __forceinline__ float* pose_joint(const PoseGrad& p, int b, int t, int w)
{
    return t == 0 ? p.root + (size_t)b * p.root_stride : p.body + (size_t)b * p.body_stride + (size_t)(t - 1) * w;
}

// ------------------------------------------------------------------------------------ forward
// One block (64 threads) per body.
__global__ __launch_bounds__(64) void pose_kernel(
    const float* __restrict__ betas, PoseRef pose, int pose2rot,
    const float* __restrict__ J_template, const float* __restrict__ J_shapedirs,
    const int32_t* __restrict__ parents, int max_depth,      // parents [24] | depths [24]
    float* __restrict__ R_out,      // [B,24,9]
    float* __restrict__ J_out,      // [B,24,3]
    float* __restrict__ world_out,  // [B,24,12] world rotation | world translation (= posed joint)
    float* __restrict__ A_out,      // [B,24,12] rotation | translation relative to the rest pose
    float* __restrict__ feat, int fpad)       // [224][fpad]: feature-major, bodies contiguous (blend_kernel)
{
    __shared__ float sR[kJoints][9];
    __shared__ float sJ[kJoints][3];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* be = betas + (size_t)b * kBetas;
    if (t < kJoints) {
        M3 r;
        if (pose2rot) r = rodrigues(pose_joint(pose, b, t, 3));
        else {
            const float* src = pose_joint(pose, b, t, 9);
#pragma unroll
            for (int e = 0; e < 9; ++e) r.m[e] = src[e];
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            sR[t][e] = r.m[e];
            R_out[((size_t)b * kJoints + t) * 9 + e] = r.m[e];
            if (t > 0) feat[(size_t)((t - 1) * 9 + e) * fpad + b] = r.m[e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
        }
    }
    for (int i = t; i < kJoints * 3; i += 64) {
        float acc = J_template[i];
#pragma unroll
        for (int l = 0; l < kBetas; ++l) acc += J_shapedirs[i * kBetas + l] * be[l];
        sJ[i / 3][i % 3] = acc;
        J_out[(size_t)b * kJoints * 3 + i] = acc;
    }
    if (t < kBetas) feat[(size_t)(kPoseFeat + t) * fpad + b] = be[t];
    if (t >= 32 && t < 32 + kFeatRows - 217) feat[(size_t)(217 + t - 32) * fpad + b] = t == 32 ? 1.0f : 0.0f;
    // Kinematic chain, one tree level at a time (SMPL: 9 levels), every joint of a level on its own lane, the world
    // transforms in LDS.  (One lane walking all 24 joints through global memory was 23 store -> load round trips.)
    __shared__ float sW[kJoints][12];
    const int parent = t < kJoints ? parents[t] : -1, depth = t < kJoints ? parents[kJoints + t] : -1;
    __syncthreads();
    for (int level = 0; level <= max_depth; ++level) {
        if (t < kJoints && depth == level) {
            const int k = t;
            M3 rk;
#pragma unroll
            for (int e = 0; e < 9; ++e) rk.m[e] = sR[k][e];
            M3 rw;
            float tw[3];
            if (k == 0) {
                rw = rk;
                tw[0] = sJ[0][0]; tw[1] = sJ[0][1]; tw[2] = sJ[0][2];
            } else {
                const int p = parent;
                M3 rp;
#pragma unroll
                for (int e = 0; e < 9; ++e) rp.m[e] = sW[p][e];
                rw = mul(rp, rk);
                const float rel[3] = {sJ[k][0] - sJ[p][0], sJ[k][1] - sJ[p][1], sJ[k][2] - sJ[p][2]};
                mulv(rp, rel, tw);
                tw[0] += sW[p][9]; tw[1] += sW[p][10]; tw[2] += sW[p][11];
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) sW[k][e] = rw.m[e];
#pragma unroll
            for (int c = 0; c < 3; ++c) sW[k][9 + c] = tw[c];
        }
        __syncthreads();
    }
    if (t < kJoints) {
        const int k = t;
        float* W = world_out + ((size_t)b * kJoints + k) * 12;
        float* A = A_out + ((size_t)b * kJoints + k) * 12;
        M3 rw;
#pragma unroll
        for (int e = 0; e < 9; ++e) rw.m[e] = sW[k][e];
        float rj[3];
        mulv(rw, sJ[k], rj);
#pragma unroll
        for (int e = 0; e < 9; ++e) { W[e] = rw.m[e]; A[e] = rw.m[e]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) { W[9 + c] = sW[k][9 + c]; A[9 + c] = sW[k][9 + c] - rj[c]; }
    }
}

// v_posed[b][n] = sum_k feat[k][b] * blend[k][n],  feat = [R[1:] - I (207) | beta (10) | 1 | 0 ...] (pose_kernel).
// One WAVEFRONT per 16 bodies x 16 columns, the whole K = 224 in one chain of 56 v_mfma_f32_16x16x4_f32 (exact f32 FMA
// chains, one fixed order): no split of K, so no partial tiles, no LDS, no barrier -- and every one of a wavefront's 112
// operand loads (4 bytes per lane: 110 registers) is in flight before its first MFMA waits.
// What bounds it: NOT the stream of the matrix -- 18.6 MB arrive in 2.5 - 4 us even from a cold cache (tools/ubench/
// stream_read.hip; the four body tiles of a batch of 64 read it four times, out of the L2 / the Infinity Cache) -- but the
// FP32 matrix cores: 4 x 1296 x 56 MFMAs of 32 cycles each over 1024 SIMDs are 3.8 us at batch 64 with every SIMD evenly
// loaded (5184 wavefronts: 5 or 6 per SIMD); a batch of 8 is one body tile, a quarter of that (round 4: always four).
// Measured on the way (batch 64): K split over the four wavefronts of a workgroup with 4 x 2 tiles each (round 4) 14.5 us,
// with 4 x 4 tiles and 16-byte operand loads 20 us (half as many wavefronts, twice as long), with the features formed
// in-kernel (no wait for pose_kernel, Rodrigues of the quarter's joints per wavefront) 17 us.
constexpr int kBlendSteps = kFeatRows / 4;              // 56 MFMA k-steps (K = 4 each)
constexpr int kBlendCols = 64;                          // columns per workgroup: four wavefronts, 16 each
template <int kTiles>                                   // body tiles per wavefront (row i of tile t = body kTiles i + t)
__global__ __launch_bounds__(256) void blend_kernel(
    const float* __restrict__ feat, int fpad, const float* __restrict__ blend, int B, int N3p, float* __restrict__ v_posed)
{
    typedef float avec __attribute__((ext_vector_type(kTiles)));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lq = lane >> 4;
    const int m_base = blockIdx.y * (16 * kTiles);
    const int col = blockIdx.x * kBlendCols + wave * 16 + lm;
    const float* b_ptr = blend + (size_t)lq * N3p + col;
    const float* a_ptr = feat + (size_t)lq * fpad + m_base + kTiles * lm;
    float bv[kBlendSteps];
    avec av[kBlendSteps];
#pragma unroll
    for (int i = 0; i < kBlendSteps; ++i) {
        bv[i] = b_ptr[(size_t)(4 * i) * N3p];
        av[i] = *(const avec*)(a_ptr + (size_t)(4 * i) * fpad);
    }
    __builtin_amdgcn_sched_barrier(0);          // everything is in flight before the first MFMA waits
    f32x4 acc[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kBlendSteps; ++i)
#pragma unroll
        for (int t = 0; t < kTiles; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][t], bv[i], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m_base + (lq * 4 + r) * kTiles + t;
            if (m < B) v_posed[(size_t)m * N3p + col] = acc[t][r];             // (padding columns: zeros)
        }
}

// The form for more than 32 bodies (round 4's kernel on the padded matrix): 64 bodies x 32 columns per workgroup, its four
// wavefronts take a quarter of K each and hold all 4 x 2 tiles, the partial tiles meet in LDS.  At batch 64 every form is
// bound by what its wavefronts ask the L1 for, not by the matrix cores or the 18.6 MB of the matrix (tools/ubench/
// blend_phases.hip: clock stamps per wavefront): one tile per wavefront re-reads the features and the matrix four times
// (148 MB through 64 B / clk / CU: operands arrive after a median 11 k cycles), four body tiles per wavefront 91 MB (4.9 k
// cycles, then 2 x 7.7 k cycles of MFMAs on the SIMDs that hold two of the 1296 wavefronts), this one 54 MB (7.1 k cycles,
// 3.9 k of MFMAs, 5.7 k for the exchange + stores): 12.3 us back to back against 14.2 - 15.0.
constexpr int kBlendWaveSteps = kBlendSteps / 4;        // per wavefront (K split four ways)
constexpr int kBlendJ = 2;                              // 16-column tiles per workgroup (K-split form)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void blend_ksplit_kernel(
    const float* __restrict__ feat, int fpad, const float* __restrict__ blend, int B, int N3p,
    float* __restrict__ v_posed)
{
    __shared__ float red[4][3][kBlendJ * 4][64];     // [body tile][sender slot][j * 4 + r][lane]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lq = lane >> 4;
    const int m_base = blockIdx.y * 64;
    const int cbase = blockIdx.x * (16 * kBlendJ) + lm * kBlendJ;      // this lane's columns: cbase + j
    const float* a_ptr = feat + m_base + lm * 4;
    const float* b_ptr = blend + cbase;                                // (rows start on 256 bytes: aligned 8-byte loads)
    f32x4 a[kBlendWaveSteps];
    f32x2 bv[kBlendWaveSteps];
#pragma unroll
    for (int i = 0; i < kBlendWaveSteps; ++i) {
        const int k = (wave * kBlendWaveSteps + i) * 4 + lq;           // (rows 220 .. 223 of the matrix are zero)
        a[i] = *(const f32x4*)(a_ptr + (size_t)k * fpad);
        bv[i] = *(const f32x2*)(b_ptr + (size_t)k * N3p);
    }
    __builtin_amdgcn_sched_barrier(0);          // the whole quarter is in flight before the first MFMA waits
    f32x4 acc[4][kBlendJ];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < kBlendJ; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kBlendWaveSteps; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float av = a[i][t];
#pragma unroll
            for (int j = 0; j < kBlendJ; ++j)
                acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[i][j], acc[t][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t == wave) continue;
        const int slot = wave < t ? wave : wave - 1;
#pragma unroll
        for (int j = 0; j < kBlendJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[t][slot][j * 4 + r][lane] = acc[t][j][r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m_base + (lq * 4 + r) * 4 + wave;              // output row lq * 4 + r of body tile `wave`
#pragma unroll
        for (int j = 0; j < kBlendJ; ++j) {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) sum = (t == wave) ? acc[t][j][r] : sum;     // own partial, no dynamic index
#pragma unroll
            for (int slot = 0; slot < 3; ++slot) sum += red[wave][slot][j * 4 + r][lane];
            if (m < B) v_posed[(size_t)m * N3p + cbase + j] = sum;              // (padding columns: zeros)
        }
    }
}

// One level of the transposing wavefront reduction: lanes whose bit `2*kHalf` differs swap halves of their value set.
template <int kHalf>
__device__ __forceinline__ void butterfly_level(float (&val)[32], int lane)
{
    constexpr int bit = 2 * kHalf;
    const bool upper = (lane & bit) != 0;
#pragma unroll
    for (int k = 0; k < kHalf; ++k) {
        const float send = upper ? val[k] : val[k + kHalf];
        const float keep = upper ? val[k + kHalf] : val[k];
        val[k] = keep + __shfl_xor(send, bit);
    }
}

// verts[b][v] = (sum_j W[v][j] A[b][j]) [v_posed[b][v]; 1]
// Also the extra-joint regression J_regressor_extra[9,V] x verts (models/smpl.py:47-48): every workgroup leaves the
// contribution of its 256 vertices, xpart[b][block][9][3] (a transposing butterfly over the wavefront, then the four
// wavefronts through LDS: a fixed order); assemble_joints_kernel adds the blocks up.  As a kernel of its own (MFMA split-K over the finished vertices)
// the regression was 21 us of pure latency in front of everything that waits for the joints.
// kSparse: every vertex has at most four non-zero weights (SMPL's own have): T = sum over those joints in ascending order --
// the same fma chain as the dense loop, whose other 20 terms are fma(0, A, T) = T, so the same BITS for finite transforms --
// with the body's 24 transforms in LDS (per-lane joints: the dense form reads them through scalar loads): 48 fma + 12
// 16-byte LDS reads per vertex instead of 288 fma + 24 weight loads.
__device__ __forceinline__ void stage_transforms(float* sA, const float* __restrict__ Ab)
{
    for (int i = threadIdx.x; i < kJoints * 12 / 4; i += kSkinBlock)
        reinterpret_cast<f32x4*>(sA)[i] = reinterpret_cast<const f32x4*>(Ab)[i];
    __syncthreads();
}
template <bool kSparse>
__global__ __launch_bounds__(kSkinBlock) void skin_kernel(
    const float* __restrict__ v_posed, int N3p, const float* __restrict__ A, const float* __restrict__ weights_t,
    const int32_t* __restrict__ skin_joint, const float* __restrict__ skin_weight,
    const float* __restrict__ Jrx, int V, float* __restrict__ verts, float* __restrict__ xpart)
{
    const int b = blockIdx.y;
    const int v = blockIdx.x * kSkinBlock + threadIdx.x;
    const bool real = v < V;
    const int vc = real ? v : V - 1;
    const float* Ab = A + (size_t)b * kJoints * 12;   // wave-uniform -> scalar loads
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    if constexpr (kSparse) {
        __shared__ __attribute__((aligned(16))) float sA[kJoints * 12];
        int jn[4];
        float wn[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { jn[n] = skin_joint[(size_t)n * V + vc]; wn[n] = skin_weight[(size_t)n * V + vc]; }
        stage_transforms(sA, Ab);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const f32x4* a = reinterpret_cast<const f32x4*>(sA + jn[n] * 12);
            const f32x4 a0 = a[0], a1 = a[1], a2 = a[2];
            const float aj[12] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3], a2[0], a2[1], a2[2], a2[3]};
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = __builtin_fmaf(wn[n], aj[e], T[e]);
        }
    } else {
        const float* w = weights_t + vc;                  // [24][V]: coalesced
#pragma unroll
        for (int j = 0; j < kJoints; ++j) {
            const float wj = w[(size_t)j * V];
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = __builtin_fmaf(wj, Ab[j * 12 + e], T[e]);
        }
    }
    const float* p = v_posed + (size_t)b * N3p + (size_t)vc * 3;
    float o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = T[3 * i] * p[0] + T[3 * i + 1] * p[1] + T[3 * i + 2] * p[2] + T[9 + i];
    if (real) {
        float* dst = verts + ((size_t)b * V + v) * 3;
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
    }
    __shared__ float red[kSkinBlock / 64][32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // 27 sums over the wavefront by a transposing butterfly: at every level a lane hands half of its values to its
    // partner and adds the partner's other half -- 16 + 8 + 4 + 2 + 1 (+ 1) shuffles instead of 27 x 6
    float val[32];
#pragma unroll
    for (int m = 0; m < kExtra; ++m) {
        const float jr = real ? Jrx[(size_t)m * V + v] : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) val[m * 3 + c] = jr * o[c];
    }
#pragma unroll
    for (int k = kExtra * 3; k < 32; ++k) val[k] = 0.f;
    butterfly_level<16>(val, lane);
    butterfly_level<8>(val, lane);
    butterfly_level<4>(val, lane);
    butterfly_level<2>(val, lane);
    butterfly_level<1>(val, lane);
    val[0] += __shfl_xor(val[0], 1);
    if ((lane & 1) == 0) red[wave][lane >> 1] = val[0];        // value index = bits 5..1 of the lane
    __syncthreads();
    if (threadIdx.x < kExtra * 3) {
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < kSkinBlock / 64; ++wv) t += red[wv][threadIdx.x];
        xpart[((size_t)b * gridDim.x + blockIdx.x) * (kExtra * 3) + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void assemble_joints_kernel(
    const float* __restrict__ world, const float* __restrict__ verts, const float* __restrict__ xpart,
    const int32_t* __restrict__ extra_ids, const int32_t* __restrict__ joint_map, int V, int skin_blocks,
    float* __restrict__ joints)   // [B,49,3]
{
    // regressed joints: 9 thread groups add every 9th block partial of the 27 values, then 27 threads add the groups
    // (one thread adding 27 partials in turn was 27 load latencies in front of everything that waits for the joints)
    constexpr int kVals = kExtra * 3, kGroups = 9;
    __shared__ float sPart[kGroups][kVals], sExtra[kVals];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < kVals * kGroups) {
        const int g = t / kVals, q = t % kVals;
        float acc = 0.f;
        for (int blk = g; blk < skin_blocks; blk += kGroups) acc += xpart[((size_t)b * skin_blocks + blk) * kVals + q];
        sPart[g][q] = acc;
    }
    __syncthreads();
    if (t < kVals) {
        float acc = 0.f;
#pragma unroll
        for (int g = 0; g < kGroups; ++g) acc += sPart[g][t];
        sExtra[t] = acc;
    }
    __syncthreads();
    if (t < kOutJoints * 3) {
        const int src = joint_map[t / 3], c = t % 3;
        float val;
        if (src < kJoints) val = world[((size_t)b * kJoints + src) * 12 + 9 + c];
        else if (src < kJoints + kPicked) val = verts[((size_t)b * V + extra_ids[src - kJoints]) * 3 + c];
        else val = sExtra[3 * (src - kJoints - kPicked) + c];
        joints[(size_t)b * kOutJoints * 3 + t] = val;
    }
}

// ----------------------------------------------------------------------------------- backward
// g_all[54][3] = scatter of g_joints [49][3] through joint_map (deterministic gather form), formed in LDS by every block
// that needs it (the skinning adjoint: picked + regressed rows; the per-body tail: the 24 chain joints) from sMap / sG the
// block loaded together with its other inputs.  As a launch of its own (joints_bwd_kernel, rounds 1-3) this was 6 us at the
// head of the backward chain for 162 additions per body.
__device__ __forceinline__ float g_all_entry(const int* sMap, const float* sG, int src, int c)
{
    float acc = 0.f;
#pragma unroll 7
    for (int o = 0; o < kOutJoints; ++o) acc += sMap[o] == src ? sG[o * 3 + c] : 0.f;
    return acc;
}

// Skinning adjoint.  Per (body, 256-vertex block):
//   g_v      = g_verts + Jrx^T g_extra + picked-vertex gradients
//   g_vposed = T_R^T g_v
//   gA_part[b][block][j][n] = sum_v W[v][j] * (g_v (x) [v_posed; 1])[n]      (MFMA, K = vertices)
template <bool kSparse>
__global__ __launch_bounds__(kSkinBlock) void skin_bwd_kernel(
    const float* __restrict__ g_verts, const float* __restrict__ g_joints, const int32_t* __restrict__ joint_map,
    const float* __restrict__ Jrx,
    const int32_t* __restrict__ extra_ids, const float* __restrict__ v_posed, int N3p, const float* __restrict__ A,
    const float* __restrict__ weights, const float* __restrict__ weights_t, const int32_t* __restrict__ skin_joint,
    const float* __restrict__ skin_weight, int V, float* __restrict__ g_vposed, float* __restrict__ gA_part,
    const long long* __restrict__ g_fixed)    // or nullptr: [B,V,3] 64-bit fixed-point sums (common.h) added to g_verts
{
    __shared__ float sG[kSkinBlock][16];      // per vertex: g_v (x) [v_posed;1], 12 used
    __shared__ int sIds[kPicked];
    __shared__ int sMap[kOutJoints];
    __shared__ float sGj[kOutJoints * 3];
    __shared__ float sGall[(kPicked + kExtra) * 3];     // rows 24..53 of g_all: picked vertices | regressed joints
    const int b = blockIdx.y;
    if (threadIdx.x < kPicked) sIds[threadIdx.x] = extra_ids[threadIdx.x];
    if (threadIdx.x < kOutJoints) sMap[threadIdx.x] = joint_map[threadIdx.x];
    if (threadIdx.x < kOutJoints * 3) sGj[threadIdx.x] = g_joints ? g_joints[(size_t)b * kOutJoints * 3 + threadIdx.x] : 0.f;
    const int v = blockIdx.x * kSkinBlock + threadIdx.x;
    const bool ok = v < V;
    const int vc = ok ? v : V - 1;
    float g[3] = {0.f, 0.f, 0.f};
    if (g_verts) {
        const float* gp = g_verts + ((size_t)b * V + vc) * 3;
        g[0] = gp[0]; g[1] = gp[1]; g[2] = gp[2];
    }
    if (g_fixed) {
        // the stage-2 tail's vertex gradient in deterministic mode, read where it is used (a conversion launch of its own was
        // 5 - 8 us of the step's serial tail); 0 + x = x: with g_verts zero this is fixed_to_float_kernel's value, bit for bit
        const long long* fp = g_fixed + ((size_t)b * V + vc) * 3;
        g[0] += fixed_value(fp[0]); g[1] += fixed_value(fp[1]); g[2] += fixed_value(fp[2]);
    }
    float jr[kExtra];
#pragma unroll
    for (int j = 0; j < kExtra; ++j) jr[j] = Jrx[(size_t)j * V + vc];
    const float* Ab = A + (size_t)b * kJoints * 12;
    float T[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) T[e] = 0.f;
    const float* p = v_posed + (size_t)b * N3p + (size_t)vc * 3;
    const float ph[4] = {p[0], p[1], p[2], 1.0f};
    if constexpr (kSparse) {                         // (see skin_kernel: the same bits as the dense loop)
        __shared__ __attribute__((aligned(16))) float sA[kJoints * 12];
        int jn[4];
        float wn[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { jn[n] = skin_joint[(size_t)n * V + vc]; wn[n] = skin_weight[(size_t)n * V + vc]; }
        stage_transforms(sA, Ab);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const f32x4* a = reinterpret_cast<const f32x4*>(sA + jn[n] * 12);
            const f32x4 a0 = a[0], a1 = a[1], a2 = a[2];
            const float aj[9] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3], a2[0]};
#pragma unroll
            for (int e = 0; e < 9; ++e) T[e] = __builtin_fmaf(wn[n], aj[e], T[e]);
        }
    } else {
        const float* w = weights_t + vc;                  // [24][V]: coalesced
#pragma unroll
        for (int j = 0; j < kJoints; ++j) {
            const float wj = w[(size_t)j * V];
#pragma unroll
            for (int e = 0; e < 9; ++e) T[e] = __builtin_fmaf(wj, Ab[j * 12 + e], T[e]);
        }
        __syncthreads();
    }
    if (threadIdx.x < (kPicked + kExtra) * 3)
        sGall[threadIdx.x] = g_all_entry(sMap, sGj, kJoints + threadIdx.x / 3, threadIdx.x % 3);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kExtra; ++j) {
        const float* gj = sGall + (kPicked + j) * 3;
        g[0] = __builtin_fmaf(jr[j], gj[0], g[0]); g[1] = __builtin_fmaf(jr[j], gj[1], g[1]); g[2] = __builtin_fmaf(jr[j], gj[2], g[2]);
    }
#pragma unroll
    for (int e = 0; e < kPicked; ++e)
        if (sIds[e] == vc) {
            const float* gj = sGall + e * 3;
            g[0] += gj[0]; g[1] += gj[1]; g[2] += gj[2];
        }
    if (!ok) { g[0] = 0.f; g[1] = 0.f; g[2] = 0.f; }
    if (ok) {
        float* o = g_vposed + (size_t)b * N3p + (size_t)v * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) o[i] = T[i] * g[0] + T[3 + i] * g[1] + T[6 + i] * g[2];
    }
    // G[n]: n = 3*i + c for the rotation part (i row of A, c column), 9 + i for the translation
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) sG[threadIdx.x][3 * i + c] = g[i] * ph[c];
        sG[threadIdx.x][9 + i] = g[i];
    }
#pragma unroll
    for (int n = 12; n < 16; ++n) sG[threadIdx.x][n] = 0.f;
    __syncthreads();
    // gA[j][n] += sum_v W[v][j] G[v][n]: the four wavefronts take (joints 0-15 | 16-31) x (first | second half of the
    // block's vertices).  The weights of the half (just read for T: cache hits) are all requested before the first MFMA
    // waits -- fetched one by one inside the loop they made it a chain of 64 load latencies.
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lm = lane & 15, lq = lane >> 4;
    const int j = (wave & 1) * 16 + lm, k_beg = (wave >> 1) * (kSkinBlock / 2);
    constexpr int kSteps = kSkinBlock / 2 / 4;
    float a[kSteps];
#pragma unroll
    for (int i = 0; i < kSteps; ++i) {
        const int vv = min(blockIdx.x * kSkinBlock + k_beg + i * 4 + lq, V - 1);      // rows past V: G is zero there
        a[i] = weights[(size_t)vv * kJoints + min(j, kJoints - 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kSteps; ++i)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(j < kJoints ? a[i] : 0.f, sG[k_beg + i * 4 + lq][lm], acc, 0, 0, 0);
    // The block's OWN [32][16] slice, plain stores (rounds 1-3: one accumulator per body, float atomics over the blocks --
    // it had to be cleared first and the sum depended on the order of arrival; pose_bwd_kernel now adds the ~27 slices in
    // a fixed order, bit-reproducible without a fixed-point mode).  The second half's partial meets the first in LDS.
    __syncthreads();                                    // every wavefront is done reading sG
    float* xch = &sG[0][0];                             // [2][4][64] floats
    if (wave >= 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[((wave & 1) * 4 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave < 2) {
        float* out = gA_part + (((size_t)b * gridDim.x + blockIdx.x) * 32 + wave * 16) * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(lq * 4 + r) * 16 + lm] = acc[r] + xch[(wave * 4 + r) * 64 + lane];
    }
}

// part[block][m][n] = sum_{k in the block's 256 columns} g_vposed[m][k] * blend[n][k];  n < 224.
// One workgroup per (256-wide K block, up to kBlendBwdGroups x 16 bodies, kBlendBwdTiles 16-column tiles), one
// wavefront per 64 of its columns: the blend tile is loaded once and used for all body groups of the wave; K index
// permuted so that every lane reads 4 consecutive floats (rows start on 8-byte boundaries only -- 3V is even, not a
// multiple of 4 -- so the loads are typed 4-byte aligned); all 24 loads of a wavefront are issued before its first MFMA
// waits.  The four partial results meet in LDS, wavefront w adds up and stores accumulators w and w + 4: plain stores
// into the block's own [Bpad][224] slice (no atomics, nothing to clear); pose_bwd_kernel adds the ~81 slices up.
constexpr int kBlendBwdTiles = 2;
constexpr int kBlendBwdChunk = 64;          // K per wavefront
constexpr int kBlendBwdWaves = 4;           // 3 workgroups fit a CU: all 81 x 7 of them run at once at batch 64
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <int kBlendBwdGroups>                  // 16-body groups per workgroup: 1 / 2 / 4 for batches up to 16 / 32 / more
__global__ __launch_bounds__(64 * kBlendBwdWaves) void blend_bwd_kernel(
    const float* __restrict__ g_vposed, const float* __restrict__ blend, int B, int N3, int N3p, int bpad,
    float* __restrict__ part)
{
    constexpr int kBlendBwdAccs = kBlendBwdTiles * kBlendBwdGroups;
    __shared__ float red[kBlendBwdWaves][kBlendBwdAccs * 4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lm = lane & 15, lq = lane >> 4;
    const int g0 = blockIdx.y * kBlendBwdGroups, j0 = blockIdx.z * kBlendBwdTiles;
    const int k_beg = (blockIdx.x * kBlendBwdWaves + wave) * kBlendBwdChunk;
    constexpr int kIters = kBlendBwdChunk / 16;
    // lane group q covers k0 + 4q .. k0 + 4q + 3 over the four MFMA steps: one 4-float load per row and step
    f32x4 a4[kIters][kBlendBwdGroups], b4[kIters][kBlendBwdTiles];
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const int kk = k_beg + it * 16 + 4 * lq;
        const int kc = kk < N3 ? kk : 0;
#pragma unroll
        for (int g = 0; g < kBlendBwdGroups; ++g)
            a4[it][g] = *(const f32x4*)(g_vposed + (size_t)min((g0 + g) * 16 + lm, B - 1) * N3p + kc);
#pragma unroll
        for (int j = 0; j < kBlendBwdTiles; ++j)
            b4[it][j] = *(const f32x4*)(blend + (size_t)min((j0 + j) * 16 + lm, kFeat - 1) * N3p + kc);
    }
    __builtin_amdgcn_sched_barrier(0);          // every load of the wavefront is in flight before the first MFMA waits
    f32x4 acc[kBlendBwdGroups][kBlendBwdTiles];
#pragma unroll
    for (int g = 0; g < kBlendBwdGroups; ++g)
#pragma unroll
        for (int j = 0; j < kBlendBwdTiles; ++j) acc[g][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const int kk = k_beg + it * 16 + 4 * lq;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool k_ok = kk + s < N3;
            float av[kBlendBwdGroups], bv[kBlendBwdTiles];
#pragma unroll
            for (int g = 0; g < kBlendBwdGroups; ++g) av[g] = ((g0 + g) * 16 + lm < B && k_ok) ? a4[it][g][s] : 0.f;
#pragma unroll
            for (int j = 0; j < kBlendBwdTiles; ++j) bv[j] = ((j0 + j) * 16 + lm < kFeat && k_ok) ? b4[it][j][s] : 0.f;
#pragma unroll
            for (int g = 0; g < kBlendBwdGroups; ++g)
#pragma unroll
                for (int j = 0; j < kBlendBwdTiles; ++j)
                    acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g], bv[j], acc[g][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < kBlendBwdGroups; ++g)
#pragma unroll
        for (int j = 0; j < kBlendBwdTiles; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(g * kBlendBwdTiles + j) * 4 + r][lane] = acc[g][j][r];
    __syncthreads();
    part += (size_t)blockIdx.x * bpad * 224;
#pragma unroll
    for (int a = wave; a < kBlendBwdAccs; a += kBlendBwdWaves) {         // the accumulators this wavefront finishes
        const int g = a / kBlendBwdTiles, j = a % kBlendBwdTiles;
        if ((g0 + g) * 16 >= B) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < kBlendBwdWaves; ++w) sum += red[w][a * 4 + r][lane];
            part[((size_t)((g0 + g) * 16 + lq * 4 + r)) * 224 + (j0 + j) * 16 + lm] = sum;
        }
    }
}

// Per body: reduce the partials, chain adjoint, Rodrigues adjoint, shape gradient.
// adam (optional, SMPLify-DC stage 2): torch.optim.Adam's update of THIS body's global_orient and body_pose rows applied
// right here, by the block that has just formed their gradients (csrc/adam.hip's arithmetic; the launch of its own, one
// workgroup at the very end of every iteration's chain, is gone).  step: the shared device counter -- every block reads it
// before the block that arrives last (ticket) advances it.
struct PoseAdam {
    float* root; float* body; int root_stride, body_stride;      // the parameters (axis-angle rows), updated in place
    float* m_root; float* v_root; float* m_body; float* v_body;  // exp_avg / exp_avg_sq, contiguous [B,3] / [B,69]
    float* step; int* ticket; float lr, eps, beta1, beta2;
};
__global__ __launch_bounds__(256) void pose_bwd_kernel(
    const float* __restrict__ gA_part, int skin_blocks, const float* __restrict__ feat_part, int feat_chunks,
    int bpad, const float* __restrict__ g_joints, const int32_t* __restrict__ joint_map,
    const float* __restrict__ R, const float* __restrict__ J,
    const float* __restrict__ world, PoseRef pose, int pose2rot,
    const float* __restrict__ J_shapedirs, const int32_t* __restrict__ parents, int max_depth,
    PoseGrad g_pose, float* __restrict__ g_betas, PoseAdam adam)
{
    __shared__ float sGA[kJoints][12];
    __shared__ float sGF[224];
    __shared__ float sGR[kJoints][9];
    __shared__ float sGJ[kJoints][3];
    __shared__ float sRw[kJoints][9];      // gradient w.r.t. the world rotations
    __shared__ float sTw[kJoints][3];      // ... world translations
    __shared__ float sW[kJoints][9], sRl[kJoints][9], sJb[kJoints][3];   // world / local rotations, rest joints
    __shared__ float sUp[kJoints][9], sGrel[kJoints][3];
    __shared__ float sGall[kJoints][3], sAA[kJoints][3];
    __shared__ int sParent[kJoints], sDepth[kJoints];
    constexpr int kGroups = 8, kPer = 11;            // slices of the blend adjoint: 88 per pass
    __shared__ float sPart[kGroups][224];
    const int b = blockIdx.x, t = threadIdx.x;
    // Everything the block reads from global memory is requested here, in one go: the kernel is a handful of threads
    // of arithmetic behind its loads, and every separate round of loads costs a full memory latency.
    // (gA: one [32][16] accumulator per body, the skinning adjoint adds its blocks with atomics.)
    float ga0 = 0.f, ga1 = 0.f, w_v = 0.f, r_v = 0.f, j_v = 0.f, aa_v = 0.f, jsd[3] = {0.f, 0.f, 0.f};
    int par_v = -1, dep_v = -1;
    __shared__ int sMap[kOutJoints];
    __shared__ float sGj[kOutJoints * 3];
    if (t < kOutJoints) sMap[t] = joint_map[t];
    if (t < kOutJoints * 3) sGj[t] = g_joints ? g_joints[(size_t)b * kOutJoints * 3 + t] : 0.f;
    const float adam_t = adam.step ? adam.step[0] + 1.0f : 0.f;       // read before any block can advance it (ticket below)
    {
        // the skinning adjoint's per-block slices [skin_blocks][32][16], added in block order (all loads in flight at once)
        const float* gp = gA_part + ((size_t)b * skin_blocks * 32 + t / 12) * 16 + t % 12;
        const float* gq = gA_part + ((size_t)b * skin_blocks * 32 + (t + 256) / 12) * 16 + (t + 256) % 12;
        const bool second = t + 256 < kJoints * 12;
        constexpr int kInFlight = 27;            // a round of loads is a full memory latency: 108 slices = 4 rounds
        for (int s0 = 0; s0 < skin_blocks; s0 += kInFlight) {
            float u0[kInFlight], u1[kInFlight];
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) {
                const bool in = s0 + u < skin_blocks;
                u0[u] = in ? gp[(size_t)(s0 + u) * 512] : 0.f;
                u1[u] = in && second ? gq[(size_t)(s0 + u) * 512] : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) { ga0 += u0[u]; ga1 += u1[u]; }
        }
    }
    if (t < kJoints * 9) {
        w_v = world[((size_t)b * kJoints + t / 9) * 12 + t % 9];
        r_v = R[(size_t)b * kJoints * 9 + t];
    }
    if (t < kJoints * 3) {
        j_v = J[(size_t)b * kJoints * 3 + t];
        if (pose2rot) aa_v = pose_joint(pose, b, t / 3, 3)[t % 3];
    }
    if (t < kJoints) { par_v = parents[t]; dep_v = parents[kJoints + t]; }
    // what the last step needs of this joint's parameters -- the prior's share of the gradient, Adam's moments and the
    // parameters themselves -- is requested here too (behind the chain each was one more memory latency)
    float add_v[3] = {0.f, 0.f, 0.f}, p_v[3] = {0.f, 0.f, 0.f}, m_v[3] = {0.f, 0.f, 0.f}, v_v[3] = {0.f, 0.f, 0.f};
    float *adam_p = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    if (pose2rot && t < kJoints) {
        if (t > 0 && g_pose.body_add) {
            const float* add = g_pose.body_add + (size_t)b * g_pose.body_add_stride + (size_t)(t - 1) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) add_v[c] = add[c];
        }
        if (adam.step) {
            adam_p = t == 0 ? adam.root + (size_t)b * adam.root_stride : adam.body + (size_t)b * adam.body_stride + (size_t)(t - 1) * 3;
            adam_m = t == 0 ? adam.m_root + (size_t)b * 3 : adam.m_body + ((size_t)b * (kJoints - 1) + (t - 1)) * 3;
            adam_v = t == 0 ? adam.v_root + (size_t)b * 3 : adam.v_body + ((size_t)b * (kJoints - 1) + (t - 1)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) { p_v[c] = adam_p[c]; m_v[c] = adam_m[c]; v_v[c] = adam_v[c]; }
        }
    }
    if (t < kBetas * kJoints) {
#pragma unroll
        for (int c = 0; c < 3; ++c) jsd[c] = J_shapedirs[((t % kJoints) * 3 + c) * kBetas + t / kJoints];
    }
    // Sum of the blend adjoint's K-block slices: 8 thread groups take every 8th slice for 32 columns at a time
    // (one thread per column adding 41 slices in turn was 41 load latencies).
    {
        const int cg = t >> 5, il = t & 31;
        const size_t slice = (size_t)bpad * 224;
        float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < feat_chunks; c0 += kGroups * kPer) {
            float v[7][kPer];
#pragma unroll
            for (int ii = 0; ii < 7; ++ii)
#pragma unroll
                for (int u = 0; u < kPer; ++u) {
                    const int c = c0 + u * kGroups + cg;
                    v[ii][u] = c < feat_chunks ? feat_part[(size_t)c * slice + (size_t)b * 224 + ii * 32 + il] : 0.f;
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < 7; ++ii)
#pragma unroll
                for (int u = 0; u < kPer; ++u) acc[ii] += v[ii][u];
        }
#pragma unroll
        for (int ii = 0; ii < 7; ++ii) sPart[cg][ii * 32 + il] = acc[ii];
    }
    sGA[t / 12][t % 12] = ga0;                                   // 256 < 288: always a valid slot
    if (t + 256 < kJoints * 12) sGA[(t + 256) / 12][(t + 256) % 12] = ga1;
    if (t < kJoints * 9) { sW[t / 9][t % 9] = w_v; sRl[t / 9][t % 9] = r_v; }
    if (t < kJoints * 3) { sJb[t / 3][t % 3] = j_v; sAA[t / 3][t % 3] = aa_v; }
    if (t < kJoints) { sParent[t] = par_v; sDepth[t] = dep_v; }
    __syncthreads();
    if (t < 224) {
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < kGroups; ++g) sum += sPart[g][t];
        sGF[t] = sum;
    }
    // rows 0..23 of g_all (the chain joints' share of the joint gradient), formed here instead of by a launch of its own
    if (t < kJoints * 3) sGall[t / 3][t % 3] = g_all_entry(sMap, sGj, t / 3, t % 3);
    __syncthreads();
    // Chain adjoint.  Phase A (one thread per joint): contributions of A_k and of the posed joint.
    // Phase B: from the leaves to the root one tree level at a time, the state in LDS: the joints of a level turn their
    // finished gradients into what their parent receives, then every parent adds its children up in a fixed order.
    // (One lane walking all 23 joints with the world transforms in global memory was 23 dependent load rounds.)
    if (t < kJoints) {
        const int k = t;
        const float* ga = sGA[k];                      // A_k = [Rw | tw - Rw J]
        const float gt[3] = {ga[9], ga[10], ga[11]};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            sTw[k][i] = sGall[k][i] + gt[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) sRw[k][3 * i + c] = ga[3 * i + c] - gt[i] * sJb[k][c];
        }
        M3 rw;
#pragma unroll
        for (int e = 0; e < 9; ++e) rw.m[e] = sW[k][e];
        float tmp[3];
        mulv_t(rw, gt, tmp);
#pragma unroll
        for (int c = 0; c < 3; ++c) sGJ[k][c] = -tmp[c];
    }
    __syncthreads();
    // one thread per (joint, matrix element): a level is two short steps instead of 24 lanes doing 3x3 products alone
    const int ek = t / 9, ee = t % 9, ei = ee / 3, ec = ee % 3;
    const bool elem = t < kJoints * 9;
    const int edepth = elem ? sDepth[ek] : -1, eparent = elem ? sParent[ek] : 0;
    unsigned kids = 0;                           // children of joint ek
    if (elem)
        for (int k = 1; k < kJoints; ++k) kids |= sParent[k] == ek ? 1u << k : 0u;
    for (int level = max_depth; level >= 1; --level) {
        if (edepth == level) {                   // this joint's gradients are complete
            const int k = ek, p = eparent;
            const float* g = sRw[k];
            const float* rk = sRl[k];
            const float* rp = sW[p];
            const float gk[3] = {sTw[k][0], sTw[k][1], sTw[k][2]};
            // (gRw_k R_k^T)[i][c] + gk[i] rel[c]  ->  what the parent's world rotation receives
            sUp[k][ee] = g[3 * ei] * rk[3 * ec] + g[3 * ei + 1] * rk[3 * ec + 1] + g[3 * ei + 2] * rk[3 * ec + 2] +
                         gk[ei] * (sJb[k][ec] - sJb[p][ec]);
            // (Rw_p^T gRw_k)[i][c]  ->  gradient of the local rotation
            sGR[k][ee] = rp[ei] * g[ec] + rp[3 + ei] * g[3 + ec] + rp[6 + ei] * g[6 + ec];
            if (ee < 3) {                        // Rw_p^T gk: the rest-joint offset
                const float grel = rp[ee] * gk[0] + rp[3 + ee] * gk[1] + rp[6 + ee] * gk[2];
                sGJ[k][ee] += grel;
                sGrel[k][ee] = grel;
            }
        }
        __syncthreads();
        if (edepth == level - 1) {               // parents of that level: children in descending order
            const int p = ek;
            float rw = sRw[p][ee], gj = ee < 3 ? sGJ[p][ee] : 0.f, tw = ee < 3 ? sTw[p][ee] : 0.f;
            for (unsigned rest = kids; rest != 0;) {
                const int k = 31 - __clz(rest);
                rest &= ~(1u << k);
                rw += sUp[k][ee];
                if (ee < 3) { gj -= sGrel[k][ee]; tw += sTw[k][ee]; }
            }
            sRw[p][ee] = rw;
            if (ee < 3) { sGJ[p][ee] = gj; sTw[p][ee] = tw; }
        }
        __syncthreads();
    }
    if (t == 0) {
#pragma unroll
        for (int e = 0; e < 9; ++e) sGR[0][e] = sRw[0][e];
#pragma unroll
        for (int c = 0; c < 3; ++c) sGJ[0][c] += sTw[0][c];
    }
    __syncthreads();
    if (t < kJoints) {
        M3 g;
#pragma unroll
        for (int e = 0; e < 9; ++e) g.m[e] = sGR[t][e] + (t > 0 ? sGF[(t - 1) * 9 + e] : 0.f);
        if (pose2rot) {
            float ga[3];
            rodrigues_bwd(sAA[t], g, ga);
            float* dst = pose_joint(g_pose, b, t, 3);
#pragma unroll
            for (int c = 0; c < 3; ++c) ga[c] += add_v[c];
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[c] = ga[c];
            if (adam.step) {                    // torch.optim.Adam's update of this joint's three parameters (adam.hip)
                const AdamScalars sc = adam_scalars(adam.lr, adam.beta1, adam.beta2, adam.eps, adam_t);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    adam_update(p_v[c], m_v[c], v_v[c], ga[c], sc);
                    adam_p[c] = p_v[c]; adam_m[c] = m_v[c]; adam_v[c] = v_v[c];
                }
            }
        } else {
            float* dst = pose_joint(g_pose, b, t, 9);
            const float* add = t > 0 && g_pose.body_add ? g_pose.body_add + (size_t)b * g_pose.body_add_stride + (size_t)(t - 1) * 9 : nullptr;
#pragma unroll
            for (int e = 0; e < 9; ++e) dst[e] = g.m[e] + (add ? add[e] : 0.0f);
        }
    }
    // shape gradient: 10 x 24 threads take one joint each, then ten add the joints up
    __shared__ float sShape[kBetas][kJoints];
    if (t < kBetas * kJoints) {
        const int l = t / kJoints, q = t % kJoints;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc += jsd[c] * sGJ[q][c];
        sShape[l][q] = acc;
    }
    __syncthreads();
    if (t < kBetas) {
        float acc = sGF[kPoseFeat + t];
        for (int q = 0; q < kJoints; ++q) acc += sShape[t][q];
        g_betas[(size_t)b * kBetas + t] = acc;
    }
    if (adam.step && t == 0) {                  // every block has read the counter by the time the last ticket is taken
        __threadfence();
        if (atomicAdd(adam.ticket, 1) == (int)gridDim.x - 1) {
            adam.step[0] = adam_t;
            *adam.ticket = 0;
        }
    }
}

struct FwdLayout { size_t R, J, world, A, feat, v_posed, partial, total; int bpad, npad, fpad; };

FwdLayout fwd_layout(const tuch_smpl_model* m, int B)
{
    FwdLayout l;
    l.bpad = ceil_div(B, 16) * 16;
    l.npad = ceil_div(3 * B, 16) * 16;
    size_t o = 0;
    l.R = o;       o += align256((size_t)B * kJoints * 9 * 4);
    l.J = o;       o += align256((size_t)B * kJoints * 3 * 4);
    l.world = o;   o += align256((size_t)B * kJoints * 12 * 4);
    l.A = o;       o += align256((size_t)B * kJoints * 12 * 4);
    l.fpad = ceil_div(B, 64) * 64;
    l.feat = o;    o += align256((size_t)l.fpad * kFeatRows * 4);
    l.v_posed = o; o += align256((size_t)B * m->N3p * 4);
    l.partial = o; o += align256((size_t)B * ceil_div(m->V, kSkinBlock) * kExtra * 3 * 4);   // xpart of skin_kernel
    l.total = o;
    return l;
}

struct BwdLayout { size_t g_vposed, gA_part, feat_part, total; int skin_blocks, feat_chunks, bpad; };

BwdLayout bwd_layout(const tuch_smpl_model* m, int B)
{
    BwdLayout l;
    l.skin_blocks = ceil_div(m->V, kSkinBlock);
    l.feat_chunks = ceil_div(m->N3, kBlendBwdChunk * kBlendBwdWaves);
    l.bpad = ceil_div(B, 16) * 16;
    size_t o = 0;
    l.g_vposed = o;  o += align256((size_t)B * m->N3p * 4);
    l.gA_part = o;   o += align256((size_t)B * l.skin_blocks * 32 * 16 * 4);   // one [32][16] slice per (body, vertex block)
    l.feat_part = o; o += align256((size_t)l.feat_chunks * l.bpad * 224 * 4);
    l.total = o;
    return l;
}

template <typename T>
int upload(T** dst, const T* src, size_t count)
{
    return tuch_table_upload((void**)dst, src, count * sizeof(T));
}

}  // namespace

extern "C" void tuch_smpl_model_destroy(tuch_smpl_model* m)
{
    if (!m) return;
    void* dev[] = {m->blend, m->J_template, m->J_shapedirs, m->weights, m->weights_t, m->Jrx, m->parents, m->extra_ids, m->joint_map,
                   m->skin_joint, m->skin_weight};
    for (void* p : dev) tuch_table_free(p);
    free(m);
}

// All arrays are HOST pointers in the layouts smplx registers them:
// v_template [V,3], shapedirs [V,3,10], posedirs [207, 3V], J_regressor [24,V], lbs_weights [V,24],
// parents [24] (parents[0] ignored), extra_vertex_ids [21], J_regressor_extra [9,V], joint_map [49].
extern "C" int tuch_smpl_model_create(tuch_smpl_model** out, int V, const float* v_template,
                                      const float* shapedirs, const float* posedirs, const float* J_regressor,
                                      const float* lbs_weights, const int32_t* parents,
                                      const int32_t* extra_vertex_ids, const float* J_regressor_extra,
                                      const int32_t* joint_map)
{
    TUCH_REQUIRE(out && V > 0 && v_template && shapedirs && posedirs && J_regressor && lbs_weights && parents &&
                     extra_vertex_ids && J_regressor_extra && joint_map, "tuch_smpl_model_create: null argument");
    for (int k = 1; k < kJoints; ++k)
        TUCH_REQUIRE(parents[k] >= 0 && parents[k] < k, "tuch_smpl_model_create: parents[%d]=%d is not a tree order", k, parents[k]);
    for (int e = 0; e < kPicked; ++e)
        TUCH_REQUIRE(extra_vertex_ids[e] >= 0 && extra_vertex_ids[e] < V, "tuch_smpl_model_create: bad picked vertex");
    for (int o = 0; o < kOutJoints; ++o)
        TUCH_REQUIRE(joint_map[o] >= 0 && joint_map[o] < kAllJoints, "tuch_smpl_model_create: bad joint_map entry");
    tuch_smpl_model* m = (tuch_smpl_model*)calloc(1, sizeof(tuch_smpl_model));
    m->V = V;
    m->N3 = 3 * V;
    m->N3p = ceil_div(m->N3, 64) * 64;
    const size_t n3 = (size_t)m->N3, n3p = (size_t)m->N3p;
    std::vector<float> blend((size_t)kFeatRows * n3p, 0.f);      // zero rows 218 .. 223, zero padding columns
    for (int k = 0; k < kPoseFeat; ++k) memcpy(blend.data() + (size_t)k * n3p, posedirs + (size_t)k * n3, sizeof(float) * n3);
    for (size_t n = 0; n < n3; ++n) {
        for (int l = 0; l < kBetas; ++l) blend[(size_t)(kPoseFeat + l) * n3p + n] = shapedirs[n * kBetas + l];
        blend[(size_t)217 * n3p + n] = v_template[n];
    }
    // joint regressor folded through the shape basis (double accumulation, rounded once)
    std::vector<float> jt(kJoints * 3), js((size_t)kJoints * 3 * kBetas);
    for (int j = 0; j < kJoints; ++j)
        for (int c = 0; c < 3; ++c) {
            double a = 0.0, s[kBetas] = {0};
            for (int v = 0; v < V; ++v) {
                const double w = J_regressor[(size_t)j * V + v];
                a += w * v_template[(size_t)v * 3 + c];
                for (int l = 0; l < kBetas; ++l) s[l] += w * shapedirs[((size_t)v * 3 + c) * kBetas + l];
            }
            jt[j * 3 + c] = (float)a;
            for (int l = 0; l < kBetas; ++l) js[(size_t)(j * 3 + c) * kBetas + l] = (float)s[l];
        }
    int32_t par[2 * kJoints];
    memcpy(par, parents, sizeof(int32_t) * kJoints);
    par[0] = -1;
    par[kJoints] = 0;
    m->max_depth = 0;
    for (int k = 1; k < kJoints; ++k) {               // parents come first: one pass
        par[kJoints + k] = par[kJoints + par[k]] + 1;
        m->max_depth = std::max(m->max_depth, (int)par[kJoints + k]);
    }
    memcpy(m->parents_host, par, sizeof(int32_t) * kJoints);
    int rc = upload(&m->blend, blend.data(), blend.size());
    if (rc == TUCH_OK) rc = upload(&m->J_template, jt.data(), jt.size());
    if (rc == TUCH_OK) rc = upload(&m->J_shapedirs, js.data(), js.size());
    if (rc == TUCH_OK) rc = upload(&m->weights, lbs_weights, (size_t)V * kJoints);
    if (rc == TUCH_OK) {
        std::vector<float> wt((size_t)kJoints * V);
        for (int v = 0; v < V; ++v)
            for (int j = 0; j < kJoints; ++j) wt[(size_t)j * V + v] = lbs_weights[(size_t)v * kJoints + j];
        rc = upload(&m->weights_t, wt.data(), wt.size());
    }
    if (rc == TUCH_OK) {
        // sparse skinning tables: the non-zero weights of every vertex in ascending joint order
        m->skin_nnz = 0;
        for (int v = 0; v < V; ++v) {
            int n = 0;
            for (int j = 0; j < kJoints; ++j) n += lbs_weights[(size_t)v * kJoints + j] != 0.0f;
            m->skin_nnz = std::max(m->skin_nnz, n);
        }
        const bool off = getenv("TUCH_SKIN_DENSE") && atoi(getenv("TUCH_SKIN_DENSE")) != 0;     // A/B, tests (read per model)
        if (m->skin_nnz <= 4 && !off) {
            std::vector<int32_t> sj((size_t)4 * V, 0);
            std::vector<float> sw((size_t)4 * V, 0.f);
            for (int v = 0; v < V; ++v) {
                int n = 0;
                for (int j = 0; j < kJoints; ++j)
                    if (lbs_weights[(size_t)v * kJoints + j] != 0.0f) {
                        sj[(size_t)n * V + v] = j;
                        sw[(size_t)n * V + v] = lbs_weights[(size_t)v * kJoints + j];
                        ++n;
                    }
            }
            rc = upload(&m->skin_joint, sj.data(), sj.size());
            if (rc == TUCH_OK) rc = upload(&m->skin_weight, sw.data(), sw.size());
        }
    }
    if (rc == TUCH_OK) rc = upload(&m->Jrx, J_regressor_extra, (size_t)kExtra * V);
    if (rc == TUCH_OK) rc = upload(&m->parents, par, 2 * kJoints);
    if (rc == TUCH_OK) rc = upload(&m->extra_ids, extra_vertex_ids, kPicked);
    if (rc == TUCH_OK) rc = upload(&m->joint_map, joint_map, kOutJoints);
    if (rc != TUCH_OK) {
        tuch_smpl_model_destroy(m);
        *out = nullptr;
        return rc;
    }
    *out = m;
    return TUCH_OK;
}

extern "C" size_t tuch_smpl_forward_workspace_bytes(const tuch_smpl_model* m, int B)
{
    return (m && B > 0) ? fwd_layout(m, B).total : 0;
}

extern "C" size_t tuch_smpl_backward_workspace_bytes(const tuch_smpl_model* m, int B)
{
    return (m && B > 0) ? bwd_layout(m, B).total : 0;
}

// global_orient: [B,3] axis-angle (pose2rot) or [B,1,3,3]; body_pose: [B,69] or [B,23,3,3] -- the two tensors of
// SMPL.forward (models/smpl.py:44-47), with row strides in floats so that views of one [B,72] pose work as well.
// The forward workspace holds the intermediates the backward pass needs and must be kept alive until then.
extern "C" int tuch_smpl_forward_split(const tuch_smpl_model* m, const float* betas, const float* global_orient,
                                       int global_orient_stride, const float* body_pose, int body_pose_stride,
                                       int pose2rot, int B, float* verts, float* joints, void* workspace,
                                       size_t workspace_bytes, void* stream)
{
    TUCH_REQUIRE(m && betas && global_orient && body_pose && verts && joints, "tuch_smpl_forward: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535, "tuch_smpl_forward: bad batch %d", B);
    const int w = pose2rot ? 3 : 9;
    TUCH_REQUIRE(global_orient_stride >= w && body_pose_stride >= 23 * w, "tuch_smpl_forward: bad pose strides %d, %d",
                 global_orient_stride, body_pose_stride);
    const FwdLayout l = fwd_layout(m, B);
    if (!workspace || workspace_bytes < l.total) {
        tuch_set_error("tuch_smpl_forward: workspace %zu < %zu bytes", workspace_bytes, l.total);
        return TUCH_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    float *R = (float*)(ws + l.R), *J = (float*)(ws + l.J), *world = (float*)(ws + l.world), *A = (float*)(ws + l.A),
          *feat = (float*)(ws + l.feat), *v_posed = (float*)(ws + l.v_posed), *partial = (float*)(ws + l.partial);
    hipStream_t s = (hipStream_t)stream;
    const PoseRef pose{global_orient, body_pose, global_orient_stride, body_pose_stride};
    hipLaunchKernelGGL(pose_kernel, dim3(B), dim3(64), 0, s, betas, pose, pose2rot, (const float*)m->J_template,
                       (const float*)m->J_shapedirs, (const int32_t*)m->parents, m->max_depth, R, J, world, A, feat, l.fpad);
    {
        // measurement switch, read once per process; anything but 1, 2 or 4 body tiles means "choose by batch size"
        static const int force = [] {
            const char* e = getenv("TUCH_BLEND_TILES");
            const int v = e ? atoi(e) : 0;
            return v == 1 || v == 2 || v == 4 ? v : 0;
        }();
        const int tiles = force ? force : (B <= 16 ? 1 : B <= 32 ? 2 : 0);
        if (tiles == 0)
            hipLaunchKernelGGL(blend_ksplit_kernel, dim3(m->N3p / (16 * kBlendJ), l.fpad / 64), dim3(256), 0, s, (const float*)feat,
                               l.fpad, (const float*)m->blend, B, m->N3p, v_posed);
        else {
            auto* kernel = tiles == 1 ? blend_kernel<1> : tiles == 2 ? blend_kernel<2> : blend_kernel<4>;
            hipLaunchKernelGGL(kernel, dim3(m->N3p / kBlendCols, ceil_div(B, 16 * tiles)), dim3(256), 0, s, (const float*)feat,
                               l.fpad, (const float*)m->blend, B, m->N3p, v_posed);
        }
    }
    hipLaunchKernelGGL(m->skin_joint ? skin_kernel<true> : skin_kernel<false>, dim3(ceil_div(m->V, kSkinBlock), B), dim3(kSkinBlock), 0, s,
                       (const float*)v_posed, m->N3p, (const float*)A, (const float*)m->weights_t, (const int32_t*)m->skin_joint,
                       (const float*)m->skin_weight, (const float*)m->Jrx, m->V, verts, partial);
    hipLaunchKernelGGL(assemble_joints_kernel, dim3(B), dim3(256), 0, s, (const float*)world, (const float*)verts,
                       (const float*)partial, (const int32_t*)m->extra_ids, (const int32_t*)m->joint_map, m->V,
                       ceil_div(m->V, kSkinBlock), joints);
    return tuch_check_launch("tuch_smpl_forward");
}

// pose: [B,72] axis-angle (pose2rot) or [B,24,3,3] rotation matrices: the concatenated form of the above.
extern "C" int tuch_smpl_forward(const tuch_smpl_model* m, const float* betas, const float* pose, int pose2rot,
                                 int B, float* verts, float* joints, void* workspace, size_t workspace_bytes,
                                 void* stream)
{
    TUCH_REQUIRE(pose, "tuch_smpl_forward: null pointer");
    const int w = pose2rot ? 3 : 9;
    return tuch_smpl_forward_split(m, betas, pose, kJoints * w, pose + w, kJoints * w, pose2rot, B, verts, joints,
                                   workspace, workspace_bytes, stream);
}

extern "C" int tuch_smpl_backward_split_add(const tuch_smpl_model* m, const float* global_orient, int global_orient_stride,
                                            const float* body_pose, int body_pose_stride, int pose2rot, int B,
                                            const void* fwd_workspace, const float* g_verts, const float* g_joints,
                                            float* g_betas, float* g_global_orient, int g_global_orient_stride,
                                            float* g_body_pose, int g_body_pose_stride, const float* g_body_pose_add,
                                            int g_body_pose_add_stride, void* workspace, size_t workspace_bytes, void* stream,
                                            const void* g_verts_fixed);

// g_verts [B,V,3] and/or g_joints [B,49,3] (either may be NULL) -> g_betas [B,10] and
// the pose gradient, written as the two tensors of tuch_smpl_forward_split (same shapes, strides in floats).
extern "C" int tuch_smpl_backward_split(const tuch_smpl_model* m, const float* global_orient, int global_orient_stride,
                                        const float* body_pose, int body_pose_stride, int pose2rot, int B,
                                        const void* fwd_workspace, const float* g_verts, const float* g_joints,
                                        float* g_betas, float* g_global_orient, int g_global_orient_stride,
                                        float* g_body_pose, int g_body_pose_stride, void* workspace,
                                        size_t workspace_bytes, void* stream)
{
    return tuch_smpl_backward_split_add(m, global_orient, global_orient_stride, body_pose, body_pose_stride, pose2rot, B,
                                        fwd_workspace, g_verts, g_joints, g_betas, g_global_orient, g_global_orient_stride,
                                        g_body_pose, g_body_pose_stride, nullptr, 0, workspace, workspace_bytes, stream, nullptr);
}

// The same with a gradient the caller already holds for body_pose (g_body_pose_add, same shape, row stride in floats; or
// NULL): g_body_pose = this call's gradient + that one -- what autograd would otherwise do in a separate add launch when
// body_pose feeds the body model AND another term (SMPLify-DC: the pose prior, losses.py:63).
static int backward_impl(const tuch_smpl_model* m, const float* global_orient, int global_orient_stride,
                         const float* body_pose, int body_pose_stride, int pose2rot, int B,
                         const void* fwd_workspace, const float* g_verts, const float* g_joints,
                         float* g_betas, float* g_global_orient, int g_global_orient_stride,
                         float* g_body_pose, int g_body_pose_stride, const float* g_body_pose_add,
                         int g_body_pose_add_stride, const PoseAdam& adam, void* workspace, size_t workspace_bytes, void* stream,
                         const void* g_verts_fixed)
{
    TUCH_REQUIRE(m && global_orient && body_pose && fwd_workspace && g_betas && g_global_orient && g_body_pose,
                 "tuch_smpl_backward: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535, "tuch_smpl_backward: bad batch %d", B);
    const int w = pose2rot ? 3 : 9;
    TUCH_REQUIRE(global_orient_stride >= w && body_pose_stride >= 23 * w && g_global_orient_stride >= w &&
                 g_body_pose_stride >= 23 * w, "tuch_smpl_backward: bad pose strides");
    const PoseRef pose{global_orient, body_pose, global_orient_stride, body_pose_stride};
    TUCH_REQUIRE(!g_body_pose_add || g_body_pose_add_stride >= 23 * w, "tuch_smpl_backward: bad stride of the added gradient");
    const PoseGrad g_pose{g_global_orient, g_body_pose, g_global_orient_stride, g_body_pose_stride, g_body_pose_add,
                          g_body_pose_add_stride};
    const FwdLayout f = fwd_layout(m, B);
    const BwdLayout l = bwd_layout(m, B);
    if (!workspace || workspace_bytes < l.total) {
        tuch_set_error("tuch_smpl_backward: workspace %zu < %zu bytes", workspace_bytes, l.total);
        return TUCH_ERR_WORKSPACE;
    }
    const char* fw = (const char*)fwd_workspace;
    const float *R = (const float*)(fw + f.R), *J = (const float*)(fw + f.J), *world = (const float*)(fw + f.world),
                *A = (const float*)(fw + f.A), *v_posed = (const float*)(fw + f.v_posed);
    char* ws = (char*)workspace;
    float *g_vposed = (float*)(ws + l.g_vposed), *gA_part = (float*)(ws + l.gA_part), *feat_part = (float*)(ws + l.feat_part);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(m->skin_joint ? skin_bwd_kernel<true> : skin_bwd_kernel<false>, dim3(l.skin_blocks, B), dim3(kSkinBlock), 0, s,
                       g_verts, g_joints, (const int32_t*)m->joint_map, (const float*)m->Jrx, (const int32_t*)m->extra_ids, v_posed,
                       m->N3p, A, (const float*)m->weights, (const float*)m->weights_t, (const int32_t*)m->skin_joint,
                       (const float*)m->skin_weight, m->V, g_vposed, gA_part, (const long long*)g_verts_fixed);
    {
        const int groups = B <= 16 ? 1 : B <= 32 ? 2 : 4;       // (the matrix cores work on whole 16-body groups: a batch of 8 = one)
        auto* kernel = groups == 1 ? blend_bwd_kernel<1> : groups == 2 ? blend_bwd_kernel<2> : blend_bwd_kernel<4>;
        hipLaunchKernelGGL(kernel, dim3(l.feat_chunks, ceil_div(l.bpad / 16, groups), 14 / kBlendBwdTiles),
                           dim3(64 * kBlendBwdWaves), 0, s, (const float*)g_vposed, (const float*)m->blend, B, m->N3, m->N3p, l.bpad, feat_part);
    }
    hipLaunchKernelGGL(pose_bwd_kernel, dim3(B), dim3(256), 0, s, (const float*)gA_part, l.skin_blocks,
                       (const float*)feat_part, l.feat_chunks, l.bpad, g_joints, (const int32_t*)m->joint_map, R, J, world, pose,
                       pose2rot, (const float*)m->J_shapedirs, (const int32_t*)m->parents, m->max_depth, g_pose, g_betas, adam);
    return tuch_check_launch("tuch_smpl_backward");
}

extern "C" int tuch_smpl_backward_split_add(const tuch_smpl_model* m, const float* global_orient, int global_orient_stride,
                                            const float* body_pose, int body_pose_stride, int pose2rot, int B,
                                            const void* fwd_workspace, const float* g_verts, const float* g_joints,
                                            float* g_betas, float* g_global_orient, int g_global_orient_stride,
                                            float* g_body_pose, int g_body_pose_stride, const float* g_body_pose_add,
                                            int g_body_pose_add_stride, void* workspace, size_t workspace_bytes, void* stream,
                                            const void* g_verts_fixed)
{
    PoseAdam none;
    memset(&none, 0, sizeof(none));
    return backward_impl(m, global_orient, global_orient_stride, body_pose, body_pose_stride, pose2rot, B, fwd_workspace,
                         g_verts, g_joints, g_betas, g_global_orient, g_global_orient_stride, g_body_pose, g_body_pose_stride,
                         g_body_pose_add, g_body_pose_add_stride, none, workspace, workspace_bytes, stream, g_verts_fixed);
}

// tuch_smpl_backward_split_add + torch.optim.Adam's update (tuch_adam_step's arithmetic) of the two pose tensors THEMSELVES,
// applied by the last backward kernel to the rows whose gradient it has just written (axis-angle poses only): for a fit
// whose optimiser holds exactly [global_orient, body_pose] and whose whole gradient arrives through this call (SMPLify-DC
// stage 2, smplifydc.py:149-183: the body model + the pose prior via g_body_pose_add).  param_*: the tensors the optimiser
// updates (normally the very memory global_orient / body_pose point to), exp_avg / exp_avg_sq contiguous [B,3] / [B,69],
// step: the optimiser's device counter (advanced by one), ticket: one zeroed int the call leaves zero.  The gradients are
// still written to g_*.
extern "C" int tuch_smpl_backward_split_adam(const tuch_smpl_model* m, const float* global_orient, int global_orient_stride,
                                             const float* body_pose, int body_pose_stride, int B,
                                             const void* fwd_workspace, const float* g_verts, const float* g_joints,
                                             float* g_betas, float* g_global_orient, int g_global_orient_stride,
                                             float* g_body_pose, int g_body_pose_stride, const float* g_body_pose_add,
                                             int g_body_pose_add_stride,
                                             float* param_global_orient, int param_global_orient_stride, float* param_body_pose,
                                             int param_body_pose_stride, float* exp_avg_global_orient, float* exp_avg_sq_global_orient,
                                             float* exp_avg_body_pose, float* exp_avg_sq_body_pose, float* step, int* ticket,
                                             float lr, float beta1, float beta2, float eps,
                                             void* workspace, size_t workspace_bytes, void* stream, const void* g_verts_fixed)
{
    TUCH_REQUIRE(param_global_orient && param_body_pose && exp_avg_global_orient && exp_avg_sq_global_orient && exp_avg_body_pose &&
                 exp_avg_sq_body_pose && step && ticket, "tuch_smpl_backward_split_adam: null pointer");
    TUCH_REQUIRE(param_global_orient_stride >= 3 && param_body_pose_stride >= 69, "tuch_smpl_backward_split_adam: bad parameter strides");
    const PoseAdam adam{param_global_orient, param_body_pose, param_global_orient_stride, param_body_pose_stride,
                        exp_avg_global_orient, exp_avg_sq_global_orient, exp_avg_body_pose, exp_avg_sq_body_pose, step, ticket,
                        lr, eps, beta1, beta2};
    return backward_impl(m, global_orient, global_orient_stride, body_pose, body_pose_stride, 1, B, fwd_workspace,
                         g_verts, g_joints, g_betas, g_global_orient, g_global_orient_stride, g_body_pose, g_body_pose_stride,
                         g_body_pose_add, g_body_pose_add_stride, adam, workspace, workspace_bytes, stream, g_verts_fixed);
}

// pose / g_pose: [B,72] or [B,24,3,3], the concatenated form.
extern "C" int tuch_smpl_backward(const tuch_smpl_model* m, const float* pose, int pose2rot, int B,
                                  const void* fwd_workspace, const float* g_verts, const float* g_joints,
                                  float* g_betas, float* g_pose, void* workspace, size_t workspace_bytes,
                                  void* stream)
{
    TUCH_REQUIRE(pose && g_pose, "tuch_smpl_backward: null pointer");
    const int w = pose2rot ? 3 : 9;
    return tuch_smpl_backward_split(m, pose, kJoints * w, pose + w, kJoints * w, pose2rot, B, fwd_workspace, g_verts,
                                    g_joints, g_betas, g_pose, kJoints * w, g_pose + w, kJoints * w, workspace,
                                    workspace_bytes, stream);
}

extern "C" int tuch_smpl_model_info(const tuch_smpl_model* m, int* V)
{
    TUCH_REQUIRE(m, "tuch_smpl_model_info: null model");
    if (V) *V = m->V;
    return TUCH_OK;
}
