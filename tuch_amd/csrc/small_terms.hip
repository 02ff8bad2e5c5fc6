// The O(B x 49) part of the SMPLify-DC objective in one kernel (K8 of SURVEY.md §2.2):
//   reprojection: pinhole projection with identity rotation (tuch/utils/geometry.py:83-111),
//                 Geman-McClure robustifier (tuch/smplify/losses.py:25-32), confidence^2 weights
//                 (losses.py:56-61)
//   pose prior:   max-mixture GMM, min_m [ 0.5 (p-mu_m)^T P_m (p-mu_m) - log w'_m ]
//                 (tuch/smplify/prior.py:117-132)
// Forward value per body and the gradients w.r.t. joints, camera translation and body pose for a
// unit upstream gradient (the caller scales them).  Replaces ~75 tiny torch kernels per step.
#include "common.h"

namespace {

constexpr int kBlock = 128;
constexpr int kPoseDim = 69;
constexpr int kMaxGauss = 16;

__global__ __launch_bounds__(kBlock) void small_terms_kernel(
    const float* __restrict__ joints,     // [B,J,3]
    const float* __restrict__ cam_t,      // [B,3]
    const float* __restrict__ cam_c,      // [B,2]
    const float* __restrict__ j2d,        // [B,J,2]
    const float* __restrict__ conf,       // [B,J]
    const float* __restrict__ pose,       // [B,69] or nullptr (no prior)
    const float* __restrict__ means,      // [M,69]
    const float* __restrict__ prec,       // [M,69,69]
    const float* __restrict__ log_w,      // [M] log(nll_weights)
    int J, int M, float focal, float sigma, float prior_scale,
    float* __restrict__ out,              // [B,2]: reprojection sum, prior_scale * prior
    float* __restrict__ g_joints,         // [B,J,3]
    float* __restrict__ g_cam,            // [B,3]
    float* __restrict__ g_pose)           // [B,69]
{
    __shared__ float red[kBlock];
    __shared__ float sdiff[kPoseDim];
    __shared__ float sq[kMaxGauss];
    __shared__ float sgc[3][kBlock];
    const int b = blockIdx.x, t = threadIdx.x;
    // ---- reprojection
    float loss = 0.f, gcx = 0.f, gcy = 0.f, gcz = 0.f;
    const float s2 = sigma * sigma;
    for (int j = t; j < J; j += kBlock) {
        const float* X = joints + ((size_t)b * J + j) * 3;
        const float x = X[0] + cam_t[3 * b], y = X[1] + cam_t[3 * b + 1], z = X[2] + cam_t[3 * b + 2];
        const float iz = 1.0f / z;
        const float rx = focal * (x * iz) + cam_c[2 * b] - j2d[((size_t)b * J + j) * 2];
        const float ry = focal * (y * iz) + cam_c[2 * b + 1] - j2d[((size_t)b * J + j) * 2 + 1];
        const float c2 = conf[(size_t)b * J + j] * conf[(size_t)b * J + j];
        const float dx = s2 + rx * rx, dy = s2 + ry * ry;
        loss += c2 * (s2 * rx * rx / dx + s2 * ry * ry / dy);
        const float gx = c2 * 2.0f * s2 * s2 * rx / (dx * dx), gy = c2 * 2.0f * s2 * s2 * ry / (dy * dy);
        const float a = gx * focal * iz, c = gy * focal * iz;
        const float e = -(gx * focal * x + gy * focal * y) * iz * iz;
        float* g = g_joints + ((size_t)b * J + j) * 3;
        g[0] = a; g[1] = c; g[2] = e;
        gcx += a; gcy += c; gcz += e;
    }
    red[t] = loss; sgc[0][t] = gcx; sgc[1][t] = gcy; sgc[2][t] = gcz;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (t < s) {
            red[t] += red[t + s];
            sgc[0][t] += sgc[0][t + s]; sgc[1][t] += sgc[1][t + s]; sgc[2][t] += sgc[2][t + s];
        }
        __syncthreads();
    }
    if (t == 0) {
        out[2 * b] = red[0];
        g_cam[3 * b] = sgc[0][0]; g_cam[3 * b + 1] = sgc[1][0]; g_cam[3 * b + 2] = sgc[2][0];
    }
    // ---- pose prior
    if (!pose) {
        if (t == 0) out[2 * b + 1] = 0.f;
        return;
    }
    int best = 0;
    float best_ll = 0.f;
    for (int m = 0; m < M; ++m) {
        __syncthreads();
        if (t < kPoseDim) sdiff[t] = pose[(size_t)b * kPoseDim + t] - means[m * kPoseDim + t];
        __syncthreads();
        float part = 0.f;
        if (t < kPoseDim) {
            const float* row = prec + ((size_t)m * kPoseDim + t) * kPoseDim;
            float acc = 0.f;
            for (int k = 0; k < kPoseDim; ++k) acc += row[k] * sdiff[k];
            part = acc * sdiff[t];
        }
        red[t] = part;
        __syncthreads();
        for (int s = kBlock / 2; s > 0; s >>= 1) {
            if (t < s) red[t] += red[t + s];
            __syncthreads();
        }
        if (t == 0) sq[m] = 0.5f * red[0] - log_w[m];
    }
    __syncthreads();
    best_ll = sq[0];
    for (int m = 1; m < M; ++m)
        if (sq[m] < best_ll) { best_ll = sq[m]; best = m; }       // first minimum, as torch.min
    if (t == 0) out[2 * b + 1] = prior_scale * best_ll;
    if (t < kPoseDim) sdiff[t] = pose[(size_t)b * kPoseDim + t] - means[best * kPoseDim + t];
    __syncthreads();
    if (t < kPoseDim) {
        // d/dp [0.5 d^T P d] = 0.5 (P + P^T) d
        const float* P = prec + (size_t)best * kPoseDim * kPoseDim;
        float acc = 0.f;
        for (int k = 0; k < kPoseDim; ++k) acc += (P[t * kPoseDim + k] + P[k * kPoseDim + t]) * sdiff[k];
        g_pose[(size_t)b * kPoseDim + t] = prior_scale * 0.5f * acc;
    }
}

}  // namespace

extern "C" int tuch_smplify_small_terms(const float* joints, const float* camera_t, const float* camera_center,
                                        const float* joints_2d, const float* joints_conf, const float* body_pose,
                                        const float* gmm_means, const float* gmm_precisions,
                                        const float* gmm_log_weights, int B, int num_joints, int num_gaussians,
                                        float focal_length, float sigma, float prior_scale, float* out,
                                        float* grad_joints, float* grad_camera_t, float* grad_body_pose,
                                        void* stream)
{
    TUCH_REQUIRE(joints && camera_t && camera_center && joints_2d && joints_conf && out && grad_joints &&
                     grad_camera_t, "tuch_smplify_small_terms: null pointer");
    TUCH_REQUIRE(B > 0 && num_joints > 0, "tuch_smplify_small_terms: bad sizes");
    TUCH_REQUIRE(!body_pose || (gmm_means && gmm_precisions && gmm_log_weights && grad_body_pose &&
                                num_gaussians > 0 && num_gaussians <= kMaxGauss),
                 "tuch_smplify_small_terms: bad prior arguments");
    hipLaunchKernelGGL(small_terms_kernel, dim3(B), dim3(kBlock), 0, (hipStream_t)stream, joints, camera_t,
                       camera_center, joints_2d, joints_conf, body_pose, gmm_means, gmm_precisions,
                       gmm_log_weights, num_joints, num_gaussians, focal_length, sigma, prior_scale, out,
                       grad_joints, grad_camera_t, grad_body_pose);
    return tuch_check_launch("tuch_smplify_small_terms");
}

// ---- the stage-1 objective of SMPLify-DC as ONE launch (tuch/smplify/losses.py:125-152: camera_fitting_loss) ----------------
//   total = sum_b [ sum_j conf^2 gmof(proj - j2d) + dw^2 (t_z - t_z^est)^2 + sw^2 |beta|^2 ]
// and its gradients w.r.t. joints, camera translation and betas for a unit upstream gradient.  One block per body; the
// block that arrives last (ticket: one int the call finds zero and leaves zero) adds the bodies' shares up in body order.
// As torch ops (projection, division, gmof, three reductions and their autograd) the loop body was ~50 launches of ~3 us
// at the end of every stage-1 iteration: a quarter of a 100 + 100-iteration fit (BASELINE config 3).
namespace {
__global__ __launch_bounds__(kBlock) void stage1_terms_kernel(
    const float* __restrict__ joints, const float* __restrict__ cam_t, const float* __restrict__ cam_t_est,
    const float* __restrict__ cam_c, const float* __restrict__ j2d, const float* __restrict__ conf,
    const float* __restrict__ betas, int J, int NB, float focal, float sigma, float depth_w2, float shape_w2,
    float* __restrict__ share, int* __restrict__ ticket, float* __restrict__ out,
    float* __restrict__ g_joints, float* __restrict__ g_cam, float* __restrict__ g_betas)
{
    __shared__ float red[kBlock];
    __shared__ float sgc[3][kBlock];
    __shared__ bool last;
    const int b = blockIdx.x, t = threadIdx.x;
    float loss = 0.f, gcx = 0.f, gcy = 0.f, gcz = 0.f;
    const float s2 = sigma * sigma;
    for (int j = t; j < J; j += kBlock) {                     // (the arithmetic of small_terms_kernel)
        const float* X = joints + ((size_t)b * J + j) * 3;
        const float x = X[0] + cam_t[3 * b], y = X[1] + cam_t[3 * b + 1], z = X[2] + cam_t[3 * b + 2];
        const float iz = 1.0f / z;
        const float rx = focal * (x * iz) + cam_c[2 * b] - j2d[((size_t)b * J + j) * 2];
        const float ry = focal * (y * iz) + cam_c[2 * b + 1] - j2d[((size_t)b * J + j) * 2 + 1];
        const float c2 = conf[(size_t)b * J + j] * conf[(size_t)b * J + j];
        const float dx = s2 + rx * rx, dy = s2 + ry * ry;
        loss += c2 * (s2 * rx * rx / dx + s2 * ry * ry / dy);
        const float gx = c2 * 2.0f * s2 * s2 * rx / (dx * dx), gy = c2 * 2.0f * s2 * s2 * ry / (dy * dy);
        const float a = gx * focal * iz, c = gy * focal * iz;
        const float e = -(gx * focal * x + gy * focal * y) * iz * iz;
        float* g = g_joints + ((size_t)b * J + j) * 3;
        g[0] = a; g[1] = c; g[2] = e;
        gcx += a; gcy += c; gcz += e;
    }
    float shape = 0.f;
    if (betas && t < NB) {
        const float be = betas[(size_t)b * NB + t];
        shape = shape_w2 * be * be;
        if (g_betas) g_betas[(size_t)b * NB + t] = 2.0f * shape_w2 * be;
    }
    red[t] = loss + shape; sgc[0][t] = gcx; sgc[1][t] = gcy; sgc[2][t] = gcz;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (t < s) {
            red[t] += red[t + s];
            sgc[0][t] += sgc[0][t + s]; sgc[1][t] += sgc[1][t + s]; sgc[2][t] += sgc[2][t + s];
        }
        __syncthreads();
    }
    if (t == 0) {
        const float dz = cam_t[3 * b + 2] - cam_t_est[3 * b + 2];
        share[b] = red[0] + depth_w2 * dz * dz;
        g_cam[3 * b] = sgc[0][0]; g_cam[3 * b + 1] = sgc[1][0]; g_cam[3 * b + 2] = sgc[2][0] + 2.0f * depth_w2 * dz;
        __threadfence();
        last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    float acc = 0.f;
    for (int i = t; i < (int)gridDim.x; i += kBlock) acc += __builtin_nontemporal_load(share + i);
    red[t] = acc;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    if (t == 0) { out[0] = red[0]; *ticket = 0; }
}
}  // namespace

extern "C" int tuch_smplify_stage1_terms(const float* joints, const float* camera_t, const float* camera_t_est,
                                         const float* camera_center, const float* joints_2d, const float* joints_conf,
                                         const float* betas, int B, int num_joints, int num_betas, float focal_length,
                                         float sigma, float depth_weight, float shape_weight, float* share, int* ticket,
                                         float* out, float* grad_joints, float* grad_camera_t, float* grad_betas,
                                         void* stream)
{
    TUCH_REQUIRE(joints && camera_t && camera_t_est && camera_center && joints_2d && joints_conf && share && ticket && out &&
                     grad_joints && grad_camera_t, "tuch_smplify_stage1_terms: null pointer");
    TUCH_REQUIRE(B > 0 && num_joints > 0 && num_betas >= 0 && num_betas <= kBlock && (!betas || grad_betas),
                 "tuch_smplify_stage1_terms: bad sizes");
    hipLaunchKernelGGL(stage1_terms_kernel, dim3(B), dim3(kBlock), 0, (hipStream_t)stream, joints, camera_t, camera_t_est,
                       camera_center, joints_2d, joints_conf, betas, num_joints, num_betas, focal_length, sigma,
                       depth_weight * depth_weight, shape_weight * shape_weight, share, ticket, out, grad_joints,
                       grad_camera_t, grad_betas);
    return tuch_check_launch("tuch_smplify_stage1_terms");
}

// ---- SMPLify-DC objective assembly, tuch/smplify/losses.py:120-123 ---------------------------
//   total = sum_b [ reprojection_b + prior_b + 10 * (interior_b + exterior_b) + clw * sum_p r2r[b,p] ]
// one block, fixed-order tree reduction (deterministic).
namespace {
constexpr int kObjBlock = 1024;
__global__ __launch_bounds__(kObjBlock) void objective_kernel(
    const float* __restrict__ small, const float* __restrict__ terms, const float* __restrict__ r2r,
    int B, int P, float contact_scale, float r2r_scale, float* __restrict__ out)
{
    __shared__ float red[kObjBlock];
    float acc = 0.f;
    for (int i = threadIdx.x; i < 2 * B; i += kObjBlock) acc += small[i] + contact_scale * terms[i];
    if (r2r) {
        // four independent loads per trip (the single workgroup is latency-bound); fixed order per thread
        const int n = B * P;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int i = threadIdx.x;
        for (; i + 3 * kObjBlock < n; i += 4 * kObjBlock) {
            a0 += r2r[i]; a1 += r2r[i + kObjBlock]; a2 += r2r[i + 2 * kObjBlock]; a3 += r2r[i + 3 * kObjBlock];
        }
        for (; i < n; i += kObjBlock) a0 += r2r[i];
        acc += r2r_scale * ((a0 + a1) + (a2 + a3));
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kObjBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// gradients of the scalar w.r.t. its three inputs: constants times the upstream gradient
__global__ __launch_bounds__(256) void objective_bwd_kernel(
    const float* __restrict__ gout, int B, int P, float contact_scale, float r2r_scale,
    float* __restrict__ g_small, float* __restrict__ g_terms, float* __restrict__ g_r2r)
{
    const float g = gout[0];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * B) { g_small[i] = g; g_terms[i] = contact_scale * g; }
    if (g_r2r && i < B * P) g_r2r[i] = r2r_scale * g;
}

// Backward of the whole tail of the stage-2 objective in one launch: the upstream scalar times the objective's constants
// -> weights of the contact sums (zero for bodies without contact terms) and of the region minima, and the unit gradients
// the forward of small_terms_kernel left (joints, camera, pose) scaled by it.
__global__ __launch_bounds__(256) void tail_bwd_kernel(
    const float* __restrict__ gout, const uint8_t* __restrict__ valid, const float* __restrict__ gj,
    const float* __restrict__ gc, const float* __restrict__ gp, int B, int NJ, int P, float contact_scale,
    float r2r_scale, float* __restrict__ g_terms, float* __restrict__ g_r2r, float* __restrict__ gj_out,
    float* __restrict__ gc_out, float* __restrict__ gp_out)
{
    const float g = gout[0];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * B) g_terms[i] = (!valid || valid[i >> 1]) ? contact_scale * g : 0.0f;
    if (g_r2r && i < B * P) g_r2r[i] = r2r_scale * g;
    if (i < B * NJ * 3) gj_out[i] = g * gj[i];
    if (i < B * 3) gc_out[i] = g * gc[i];
    if (gp && i < B * 69) gp_out[i] = g * gp[i];
}
}  // namespace

extern "C" int tuch_smplify_tail_bwd(const float* grad_out, const uint8_t* valid, const float* gj, const float* gc,
                                     const float* gp, int B, int NJ, int P, float contact_scale, float r2r_scale,
                                     float* grad_contact, float* grad_r2r, float* gj_out, float* gc_out, float* gp_out,
                                     void* stream)
{
    TUCH_REQUIRE(grad_out && gj && gc && grad_contact && gj_out && gc_out && B > 0 && NJ > 0 && P >= 0 && (!gp || gp_out),
                 "tuch_smplify_tail_bwd: bad arguments");
    int n = B * NJ * 3;
    if (B * 69 > n) n = B * 69;
    if (B * P > n) n = B * P;
    hipLaunchKernelGGL(tail_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, grad_out, valid, gj, gc,
                       gp, B, NJ, P, contact_scale, r2r_scale, grad_contact, (P > 0 ? grad_r2r : (float*)nullptr), gj_out,
                       gc_out, gp_out);
    return tuch_check_launch("tuch_smplify_tail_bwd");
}

extern "C" int tuch_smplify_objective(const float* small_terms, const float* contact_terms, const float* r2r,
                                      int B, int P, float contact_scale, float r2r_scale, float* out,
                                      void* stream)
{
    TUCH_REQUIRE(small_terms && contact_terms && out && B > 0 && P >= 0, "tuch_smplify_objective: bad arguments");
    hipLaunchKernelGGL(objective_kernel, dim3(1), dim3(kObjBlock), 0, (hipStream_t)stream, small_terms, contact_terms,
                       (P > 0 ? r2r : (const float*)nullptr), B, P, contact_scale, r2r_scale, out);
    return tuch_check_launch("tuch_smplify_objective");
}

extern "C" int tuch_smplify_objective_bwd(const float* grad_out, int B, int P, float contact_scale,
                                          float r2r_scale, float* grad_small, float* grad_contact,
                                          float* grad_r2r, void* stream)
{
    TUCH_REQUIRE(grad_out && grad_small && grad_contact && B > 0 && P >= 0,
                 "tuch_smplify_objective_bwd: bad arguments");
    const int n = (2 * B > B * P ? 2 * B : B * P);
    hipLaunchKernelGGL(objective_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, grad_out,
                       B, P, contact_scale, r2r_scale, grad_small, grad_contact,
                       (P > 0 ? grad_r2r : (float*)nullptr));
    return tuch_check_launch("tuch_smplify_objective_bwd");
}
