/*
 * libtuch_amd -- MI355X (gfx950) implementation of TUCH's self-contact path.
 *
 * C ABI: plain device pointers and sizes, one HIP stream per call, no torch types.
 * The reference (muelea/tuch) is pure Python with no FFI layer; every entry point
 * below replaces the reference function(s) cited next to it and is what a binding
 * for that function would call (INTEGRATION.md shows the ctypes stubs).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless marked "host"; float = IEEE binary32;
 *     tensors are dense row-major with the shape given in the comment
 *   - return 0 on success, negative on failure; tuch_last_error() describes the last
 *     failure on the calling thread; nothing is thrown across the ABI
 *   - hot calls never allocate, never synchronise and are hipGraph-capturable; scratch
 *     memory is passed in by the caller (size from the matching *_workspace_bytes)
 *   - `stream` is a hipStream_t (NULL = default stream)
 *   - results are deterministic (fixed reduction order); only gradient scatters use
 *     float atomics
 */
#ifndef TUCH_AMD_H
#define TUCH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* tuch_last_error(void);
int tuch_abi_version(void);

/* ---- tuch/utils/contact.py ------------------------------------------------------- */

/* batch_pairwise_dist(x, y, use_cuda, squared), contact.py:23-47.
 * x [B,Nx,3], y [B,Ny,3] -> P [B,Nx,Ny] = |x|^2 + |y|^2 - 2 x.y (sqrt of it if !squared).
 * Materialising form kept for API parity; the hot path uses tuch_v2v_min_masked. */
int tuch_batch_pairwise_dist(const float* x, const float* y, int B, int Nx, int Ny, int squared,
                             float* P, void* stream);

/* Adjoint of batch_pairwise_dist for callers that differentiate through the matrix (the reference does:
 * tuch/smplify/losses.py:76-78 -> 115-116, tuch/eft/loss.py:142; torch autograd through the three bmm of
 * contact.py:27-29).  grad_P [B,Nx,Ny] -> grad_x [B,Nx,3] and/or grad_y [B,Ny,3] (either may be NULL):
 * grad_x_i = 2 sum_j g_ij (x_i - y_j), grad_y_j = 2 sum_i g_ij (y_j - x_i); !squared: g / (2 sqrt(P)) first.
 * Fixed summation order (no atomics). */
int tuch_batch_pairwise_dist_bwd(const float* x, const float* y, const float* grad_P, int B, int Nx, int Ny,
                                 int squared, float* grad_x, float* grad_y, void* stream);

/* Adjoint of solid_angles / winding_numbers for callers that differentiate through them (plain torch ops in the reference,
 * contact.py:79-109,147; the reference itself only calls them under torch.no_grad()).  grad_out [B,Q,F] (adjoint of
 * tuch_solid_angles) or grad_w [B,Q] (of tuch_winding_numbers), exactly one; grad_points [B,Q,3] and / or grad_triangles
 * [B,F,3,3] (either may be NULL).  torch's conventions at the singular points: d|a|/da = 0 at a = 0, NaN where the query sits
 * on a corner (atan2 at (0,0)).  Fixed summation order. */
int tuch_solid_angles_bwd(const float* points, const float* triangles, const float* grad_out, const float* grad_w,
                          int B, int Q, int F, float* grad_points, float* grad_triangles, void* stream);

/* solid_angles(points, triangles, thresh), contact.py:49-109.
 * points [B,Q,3], triangles [B,F,3,3] -> out [B,Q,F] = 2*atan2(num, den). */
int tuch_solid_angles(const float* points, const float* triangles, int B, int Q, int F, float* out,
                      void* stream);

/* winding_numbers(points, triangles, thresh), contact.py:112-147, fused with the callers'
 * `.le(0.99)` (losses.py:82, loss.py:262,297).  w [B,Q] and/or exterior [B,Q] (1 = w <= thresh);
 * either output may be NULL.  Never materialises anything of size QxF. */
size_t tuch_winding_workspace_bytes(int B, int Q, int F);
int tuch_winding_numbers(const float* points, const float* triangles, int B, int Q, int F, float* w,
                         uint8_t* exterior, float exterior_thresh, void* workspace,
                         size_t workspace_bytes, void* stream);

/* triangles = verts[b][face_tensor[0]], losses.py:81 / loss.py:260.
 * verts [B,V,3], faces [F,3] int32 -> triangles [B,F,3,3]. */
int tuch_gather_triangles(const float* verts, const int32_t* faces, int B, int V, int F,
                          float* triangles, void* stream);

/* ---- masked nearest vertex: losses.py:76-78,92-93 / loss.py:255-257,269-270 -------- */

/* geomask (geod > geothres, smplifydc.py:65 / loss.py:71) as bits: word [w][j] holds
 * columns 64w..64w+63 of row j; tuch_geomask_words(V) words per row set (even, zero padded). */
int tuch_geomask_words(int V);
size_t tuch_geomask_bits_bytes(int V);
int tuch_pack_geomask(const uint8_t* geomask /* [V,V] bytes */, int V, uint64_t* bits, void* stream);

/* For every column i: min / argmin over rows j with geomask[j][i] of |v_i - v_j|^2
 * (first index on ties, all-masked -> (inf, 0), as torch.min / torch.argmin).
 * points [B,N,3], bits from tuch_pack_geomask or a model -> min_d2 [B,N], argmin [B,N] int32. */
size_t tuch_v2v_workspace_bytes(int B, int N);
int tuch_v2v_min_masked(const float* points, const uint64_t* geomask_bits, int B, int N, float* min_d2,
                        int32_t* argmin, void* workspace, size_t workspace_bytes, void* stream);

/* Ragged variant for the HD resampling of loss.py:284-291: body b owns points
 * offsets[b]..offsets[b+1] (device int32 [B+1]); point a inherits the mask row/column of template
 * vertex vertex_ids[a] (geovec_verts, loss.py:88).  argmin is relative to the body's first point (first index
 * among equal minima, 0 when no row is admissible).  Rows are visited in chunks of 32 that are skipped when
 * their bounding box is farther than the columns' current minima: any point order is exact, points of a body
 * that are sorted by surface patch are fast. */
size_t tuch_v2v_min_indexed_workspace_bytes(int B, int max_points_per_body);
int tuch_v2v_min_indexed(const float* points, const int32_t* vertex_ids, const int32_t* offsets,
                         const uint64_t* geomask_bits, int B, int V, int max_points_per_body, float* min_d2,
                         int32_t* argmin, void* workspace, size_t workspace_bytes, void* stream);
/* The same search on the matrix cores (what the HD branch of tuch_hd_contact_fwd runs by default): distances of 32 rows x
 * 64 columns from v_mfma_f32_32x32x2_f32 in coordinates relative to the column block, the mask as a bf16 penalty product.
 * The winner of a column is the row with the smallest 20-bit key of that distance: it may differ from
 * tuch_v2v_min_indexed's between rows whose squared distances tie within ~1e-6 relative + a few ulp of the block's
 * squared radius (the reference's own |x|^2+|y|^2-2x.y form is noisier); min_d2 is the direct-difference distance of
 * the winner.  Same arguments. */
size_t tuch_v2v_min_indexed_mfma_workspace_bytes(int B, int max_points_per_body);
int tuch_v2v_min_indexed_mfma(const float* points, const int32_t* vertex_ids, const int32_t* offsets,
                              const uint64_t* geomask_bits, int B, int V, int max_points_per_body, float* min_d2,
                              int32_t* argmin, void* workspace, size_t workspace_bytes, void* stream);

/* ---- pull/push terms: losses.py:96-105 (mode 0) / loss.py:303-315 (mode 1) ---------- */

/* terms [B,2] = (sum over interior, sum over exterior) of weight*tanh(d/scale)^2 with
 * d_i = |x_i - x_partner(i)|; body_valid [B] bytes or NULL (invalid bodies give 0). */
int tuch_contact_terms_fwd(const float* points, const int32_t* partner, const uint8_t* exterior,
                           const uint8_t* body_valid, int B, int N, int mode, float euclthres,
                           float* terms, void* stream);
/* Ragged forward (HD points, loss.py:299-315): body b owns points offsets[b]..offsets[b+1] of one
 * concatenated set [N,3]; partner holds global indices; terms [B,2]. */
int tuch_contact_terms_ragged_fwd(const float* points, const int32_t* partner, const uint8_t* exterior,
                                  const int32_t* offsets, int B, int mode, float euclthres, float* terms,
                                  void* stream);
int tuch_contact_terms_ragged_bwd(const float* points, const int32_t* partner, const uint8_t* exterior,
                                  const int32_t* body_of_point, const float* grad_scale, int N, int mode,
                                  float euclthres, float* grad_points, void* stream);
/* grad_points [B,N,3] += d(terms)/d(points) . grad_scale [B,2]; grad_points zeroed by the caller. */
int tuch_contact_terms_bwd(const float* points, const int32_t* partner, const uint8_t* exterior,
                           const float* grad_scale, int B, int N, int mode, float euclthres,
                           float* grad_points, void* stream);
/* The same scatter in deterministic mode (tuch_set_deterministic): grad_fixed_zeroed = B*N*3 zeroed 64-bit words the
 * contributions are added to as fixed-point integers (order-independent), then converted: grad_points is WRITTEN
 * (no need to clear it), bit-reproducible.  Valid range as for tuch_smplify_stage2_fused. */
int tuch_contact_terms_bwd_fixed(const float* points, const int32_t* partner, const uint8_t* exterior,
                                 const float* grad_scale, int B, int N, int mode, float euclthres,
                                 void* grad_fixed_zeroed, float* grad_points, void* stream);

/* RegressorLoss.contact_loss's last line, loss.py:317 (`loss_contact.sum() / valid_fit.sum()` over the per-body terms of
 * :272 / :315): terms [B,K] summed over the bodies with valid[b] != 0 and divided by their number, in one launch.
 * out [2] = (mean, 1 / number of valid bodies); no valid body: 0 / 0 = NaN, as the reference's division.
 * Backward: grad_terms [B,K] = upstream[0] * (valid[b] ? out[1] : 0) -- what tuch_contact_terms_bwd takes as grad_scale. */
int tuch_valid_mean_fwd(const float* terms, const uint8_t* valid, int B, int K, float* out, void* stream);
int tuch_valid_mean_bwd(const float* upstream, const float* fwd_out, const uint8_t* valid, int B, int K,
                        float* grad_terms, void* stream);

/* Reprojection + pose-prior part of the SMPLify-DC objective, losses.py:56-64 (projection
 * geometry.py:83-111 with identity rotation, gmof losses.py:25-32, max-mixture prior
 * prior.py:117-132).  out [B,2] = (sum_j conf^2 gmof, prior_scale * prior); gradients for a unit
 * upstream gradient.  body_pose may be NULL (no prior). */
int tuch_smplify_small_terms(const float* joints, const float* camera_t, const float* camera_center,
                             const float* joints_2d, const float* joints_conf, const float* body_pose,
                             const float* gmm_means, const float* gmm_precisions, const float* gmm_log_weights,
                             int B, int num_joints, int num_gaussians, float focal_length, float sigma,
                             float prior_scale, float* out, float* grad_joints, float* grad_camera_t,
                             float* grad_body_pose, void* stream);

/* The stage-1 objective of SMPLify-DC, camera_fitting_loss (tuch/smplify/losses.py:125-152), in one launch:
 * out[0] = sum_b [ sum_j conf^2 gmof(proj - j2d, sigma) + depth_weight^2 (t_z - t_z^est)^2 + shape_weight^2 |betas_b|^2 ]
 * and its gradients for a unit upstream gradient: grad_joints [B,J,3], grad_camera_t [B,3], grad_betas [B,num_betas]
 * (betas / grad_betas may be NULL: no shape term).  share: B floats of scratch; ticket: one int, zero before the first
 * call and left zero by every call (the block that arrives last adds the bodies up in body order: deterministic). */
int tuch_smplify_stage1_terms(const float* joints, const float* camera_t, const float* camera_t_est,
                              const float* camera_center, const float* joints_2d, const float* joints_conf,
                              const float* betas, int B, int num_joints, int num_betas, float focal_length, float sigma,
                              float depth_weight, float shape_weight, float* share, int* ticket, float* out,
                              float* grad_joints, float* grad_camera_t, float* grad_betas, void* stream);

/* Adam update (torch.optim.Adam without weight decay / amsgrad: tuch/smplify/smplifydc.py:117,150 optimises body pose,
 * global orientation, betas, camera translation with it) of up to 8 small tensors in ONE launch, the step counter on the
 * device (capturable).  params / grads / exp_avg / exp_avg_sq: `count` device pointers each (host arrays), sizes[k]
 * floats; betas [count][2]; step: one device float = updates so far, incremented by the call. */
int tuch_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const int* sizes, const float* betas, float* step, float lr, float eps,
                   void* stream);

/* Objective assembly of losses.py:120-123 as one deterministic reduction:
 * out[0] = sum(small_terms [B,2]) + contact_scale * sum(contact_terms [B,2]) + r2r_scale * sum(r2r [B,P]). */
int tuch_smplify_objective(const float* small_terms, const float* contact_terms, const float* r2r, int B, int P,
                           float contact_scale, float r2r_scale, float* out, void* stream);
int tuch_smplify_objective_bwd(const float* grad_out, int B, int P, float contact_scale, float r2r_scale,
                               float* grad_small, float* grad_contact, float* grad_r2r, void* stream);

/* Backward of the tail of the stage-2 objective (small terms + contact sums + region minima -> scalar) in one launch:
 * grad_contact [B,2] = contact_scale * g (0 for bodies with valid == 0), grad_r2r [B,P] = r2r_scale * g, and the unit
 * gradients left by tuch_smplify_small_terms (gj [B,NJ,3], gc [B,3], gp [B,69] or NULL) scaled by g = grad_out[0]. */
int tuch_smplify_tail_bwd(const float* grad_out, const uint8_t* valid, const float* gj, const float* gc, const float* gp,
                          int B, int NJ, int P, float contact_scale, float r2r_scale, float* grad_contact, float* grad_r2r,
                          float* gj_out, float* gc_out, float* gp_out, void* stream);

/* The same tail (tuch/smplify/losses.py:96-123) in one launch per direction: contact sums + the row of region minima per
 * body and the objective's total (the block that finishes last adds the bodies up); backward: upstream scalar -> vertex
 * gradient (contact terms and region minima, grad_points pre-zeroed) and the scaled unit gradients of the small terms.
 * share: [B] floats of scratch; ticket: one int, zero before the first call, left zero by every call. */
/* tuch_smplify_stage2_finish and -- when grad_points is given -- tuch_smplify_stage2_bwd for a UNIT upstream gradient in
 * one launch: the objective is the root of the fit's autograd graph, so its vertex gradient can be written while the sums
 * are formed.  share: tuch_smplify_stage2_fused_scratch_floats(B) floats; ticket: one int, zero before the call;
 * grad_points [B,N,3] pre-zeroed or NULL; ij [B,P,2] from tuch_region_pair_min or NULL. */
size_t tuch_smplify_stage2_fused_scratch_floats(int B);
int tuch_smplify_stage2_fused(const float* points, const int32_t* partner, const uint8_t* exterior,
                              const uint8_t* body_valid, int B, int N, int mode, float euclthres,
                              const float* small_terms, const float* r2r, const int32_t* ij, int P, float contact_scale,
                              float r2r_scale, float* share, int* ticket, float* terms, float* out, float* grad_points,
                              const tuch_contact_model* model, const void* pair_keys, void* grad_fixed_zeroed, void* stream);
/* Deterministic mode -- ON by default (TUCH_DETERMINISTIC=0 in the environment when the library is loaded, or
 * tuch_set_deterministic(0), selects float atomics: last-ulp run-to-run noise in the gradients): the gradient scatters of
 * tuch_smplify_stage2_fused (contact terms, region minima) and of tuch_smpl_backward (skinning adjoint) accumulate 64-bit
 * fixed-point numbers (2^-36) with integer atomics instead of floats -- sums that do not depend on the order of arrival,
 * so an SMPLify-DC fit reproduces bit for bit.  tuch_smplify_stage2_fused then wants grad_fixed_zeroed = B*N*3 zeroed
 * 64-bit words (NULL: float atomics, whatever the mode) and converts to grad_points with a second launch; with
 * grad_points = NULL the accumulators are the result (see g_verts_fixed of tuch_smpl_backward_split_add).
 * VALID RANGE of the fixed-point sums: |sum| < 2^27 = 1.3e8 (beyond, the unsigned 64-bit accumulator wraps silently) and
 * contributions below 2^-37 = 7e-12 round to zero.  Gradient magnitudes scale with contact_scale / r2r_scale: with the
 * reference's weights (10, 2000) and metre-scale bodies the per-vertex gradients are < 1e5. */
/* n fixed-point sums -> floats (what tuch_smplify_stage2_fused does itself when grad_points is given) */
int tuch_fixed_to_float(const void* fixed, size_t n, float* out, void* stream);
void tuch_set_deterministic(int on);
int tuch_get_deterministic(void);

/* The region-pair search of tuch_region_pair_min alone, for tuch_smplify_stage2_fused(model, pair_keys): keys [B,P]
 * 64-bit words, ZERO on entry (the caller clears them with whatever else it clears); no clearing and no finalize
 * launch.  pair_keys given: r2r / ij of tuch_smplify_stage2_fused are ignored. */
int tuch_region_pair_keys(const tuch_contact_model* model, const float* verts, int B, const uint8_t* select,
                          int use_geomask, void* keys_zeroed, void* stream);
int tuch_smplify_stage2_finish(const float* points, const int32_t* partner, const uint8_t* exterior,
                               const uint8_t* body_valid, int B, int N, int mode, float euclthres,
                               const float* small_terms, const float* r2r, int P, float contact_scale, float r2r_scale,
                               float* share, int* ticket, float* terms, float* out, void* stream);
int tuch_smplify_stage2_bwd(const float* grad_out, const uint8_t* body_valid, const float* points, const int32_t* partner,
                            const uint8_t* exterior, int B, int N, int mode, float euclthres, float contact_scale,
                            const int32_t* ij, int P, float r2r_scale, const float* gj, const float* gc, const float* gp,
                            int NJ, float* grad_points, float* gj_out, float* gc_out, float* gp_out, void* stream);

/* ---- per-model constants -------------------------------------------------------------
 * Host tables in, device copies kept by the handle.  Segments follow
 * tuch/utils/segmentation.py:29-99: seg_q = segment_vidx lists; seg_faces = faces of the
 * closed segment where cap vertex c (global numbering over all segments) has index V + c;
 * cap_* = the ordered boundary loops whose mean is the cap vertex.  Regions follow
 * ContactSigSMPL / classes (train_module.py:64-66): CSR vertex lists and [P,2] pairs. */
typedef struct tuch_contact_model tuch_contact_model;

int tuch_contact_model_create(tuch_contact_model** out, int V, int F, const int32_t* faces /* host [F,3] */,
                              const uint8_t* geomask /* host [V,V] bytes or NULL */,
                              int num_segments, const int32_t* seg_q_off, const int32_t* seg_q_vidx,
                              const int32_t* seg_f_off, const int32_t* seg_faces,
                              int num_caps, const int32_t* cap_off, const int32_t* cap_vidx,
                              int num_regions, const int32_t* region_off, const int32_t* region_vidx,
                              int num_pairs, const int32_t* pairs);
void tuch_contact_model_destroy(tuch_contact_model* model);
/* Switches of the hot calls (A/B measurements, tests): winding_ray (0 never / 1 when only flags are wanted / 2 also for
 * w), winding_tree, winding_strips, tree_waves, ray_pair_cap, ray_waves, ray_fans, v2v_tree, v2v_flat, v2v_pairs, v2v_waves, v2v_lds,
 * seg_splits, seg_fused, seg_assist (fixed at create), hd_search, hd_search_waves, hd_overlap, canary.  (Deterministic mode is process-wide: tuch_set_deterministic.)  The environment variables TUCH_<NAME> are read ONCE, by
 * tuch_contact_model_create; afterwards only set_option changes a model's switches -- no hot call looks at the
 * environment, so a captured hipGraph cannot depend on it.  (The workspace sizes depend on ray_pair_cap and canary:
 * query *_workspace_bytes again after changing them.) */
int tuch_contact_model_set_option(tuch_contact_model* model, const char* name, int value);
int tuch_contact_model_get_option(const tuch_contact_model* model, const char* name, int* value);
/* Option canary = 1 (debug): every region of every workspace of the model's hot calls (nearest-vertex search, inside
 * test incl. its pair lists, winding of points, the HD branch's worst-case buffers and saved state) is followed by 256
 * guard bytes, written with 0xDEADBEEF before the call's kernels and compared after them, on the call's stream.
 * canary_hits: guard words found changed since the last reset (synchronises the device).  selftest: arms three regions
 * in `workspace` (>= 4 KiB), overruns one by a word on purpose and returns the number of hits counted (1). */
int tuch_contact_model_canary_hits(const tuch_contact_model* model, int* hits_host, int reset);
int tuch_contact_model_canary_selftest(const tuch_contact_model* model, void* workspace, size_t workspace_bytes, void* stream);
const uint64_t* tuch_contact_model_mask_bits(const tuch_contact_model* model);
const int32_t* tuch_contact_model_faces(const tuch_contact_model* model);
/* eight device ints, zero between calls: arrival counters for tuch_smplify_stage2_finish (one per stream in flight) */
int32_t* tuch_contact_model_tickets(const tuch_contact_model* model);
/* The geodesic mask packed in the cluster tree's vertex order (device, same layout as mask_bits with vertex v
 * replaced by its position in tuch_cluster_tree_export's qperm), or NULL without tree or mask: neighbouring
 * positions are neighbours on the surface, so a wavefront of nearby points touches few mask words. */
const uint64_t* tuch_contact_model_tree_mask_bits(const tuch_contact_model* model);
int tuch_contact_model_info(const tuch_contact_model* model, int* V, int* F, int* num_segments,
                            int* seg_q_total, int* num_pairs);
/* The triangle-strip walk of the faces used by the winding kernel (inspection / tests):
 * stream_len vertex ids with sign 0 (prime) or +-1 (emit triangle of the last three ids). */
int tuch_contact_model_strips(const tuch_contact_model* model, int* stream_len, int* num_strips,
                              int32_t* vidx_host, float* sign_host);

/* Measurement aid for the hierarchical winding numbers inside tuch_exterior_flags: walks the cluster tree
 * for verts [B,V,3] and reports the stream elements the wavefronts stepped through.  out_host[4] =
 * {leaf-strip elements, cap elements, wavefronts, elements of the flat strip stream}; one element step
 * serves 64 queries (one wavefront).  Workspace as for tuch_exterior_flags.  Synchronises the stream. */
int tuch_winding_tree_work(const tuch_contact_model* model, const float* verts, int B, void* workspace,
                           size_t workspace_bytes, unsigned long long* out_host, void* stream);

/* Measurement aid for the ray-crossing inside test (csrc/ray_winding.hip) that tuch_exterior_flags and
 * tuch_winding_points use when only the flags are wanted: out_host[4] = {strip elements stepped through by all
 * wavefronts (64 queries each), (ray, element) pairs whose ray passes the slabs of the element's leaf, 64 x elements
 * listed, wavefronts}: [1] / [2] is the share of lanes that can have a crossing at all.  Workspace as for tuch_exterior_flags.  Synchronises the stream. */
int tuch_ray_work(const tuch_contact_model* model, const float* verts, int B, void* workspace,
                  size_t workspace_bytes, unsigned long long* out_host, void* stream);
/* The cluster tree the model built for itself: qperm_host [V] = vertices in tree order, face_leaf_host [F] = leaf
 * (preorder sequence number) of every face; either may be NULL.  Fails when the model has no tree. */
int tuch_contact_model_tree_order(const tuch_contact_model* model, int32_t* qperm_host, int32_t* face_leaf_host);

/* Model-level form of tuch_v2v_min_masked: uses the model's geodesic mask and, when the model has a
 * cluster tree, a pruned walk that gives the same minima (rows whose posed box is farther than a column's
 * current minimum, or that the mask rules out entirely, are skipped).  Exact ties between rows are
 * resolved deterministically (smallest row in the tree's vertex order). */
size_t tuch_v2v_model_workspace_bytes(const tuch_contact_model* model, int B);
/* hint_inout (optional, tuch_v2v_hint_bytes bytes, device, opaque, zero-initialised by the caller): the partners found
 * by this call are left there and seed the next call with the same buffer -- in an iterative fit the previous
 * iteration's partner is still admissible and almost as close, so nearly every box is pruned at once.  The result
 * does not depend on the hint (any content is safe); only the run time does.  Used only with a cluster tree. */
size_t tuch_v2v_hint_bytes(const tuch_contact_model* model, int B);
int tuch_v2v_min_model(const tuch_contact_model* model, const float* verts, int B, float* min_d2,
                       int32_t* argmin, void* hint_inout, void* workspace, size_t workspace_bytes, void* stream);
/* The same search for callers that run other kernels beside it on another stream (as ops.ContactModel.exterior_and_partner
 * does with the inside test): leave_room & 1 caps the walk's occupancy so that the neighbours' small kernels are not
 * starved of wave slots; leave_room & 2: the caller expects hint_inout to hold near-final partners (an iterative fit calling
 * again after a small parameter update) -- the search then uses fewer, longer wavefronts.  Same results either way. */
int tuch_v2v_min_model_shared(const tuch_contact_model* model, const float* verts, int B, float* min_d2,
                              int32_t* argmin, void* hint_inout, void* workspace, size_t workspace_bytes, int leave_room,
                              void* stream);
/* The same; additionally `zero` (16-byte aligned, zero_bytes a multiple of 16; or NULL / 0) is cleared by the call's first
 * kernel: on the stream, before anything the caller enqueues behind the call -- instead of a fill launch of its own
 * (SMPLify-DC stage 2: the vertex gradient its tail scatters into, the tail's arrival counter, the region pairs' keys). */
int tuch_v2v_min_model_shared_zero(const tuch_contact_model* model, const float* verts, int B, float* min_d2, int32_t* argmin,
                                   void* hint_inout, void* workspace, size_t workspace_bytes, int leave_room, void* zero,
                                   size_t zero_bytes, void* stream);

/* The search for a caller that only needs partners within `cap` of vertices OUTSIDE the body -- the SMPLify-DC contact term:
 * an exterior vertex contributes only if its nearest admissible vertex is closer than euclthres (tuch/smplify/losses.py:99-104),
 * an interior one at any distance (:100-101).  prev_exterior [B,V]: the flags of the previous iteration (tuch_v2v_min_model_fix
 * keeps them current; zeros at first: nothing capped).  A vertex with prev_exterior != 0 and an admissible hint starts its
 * search at cap^2: found closer -> exact; else min_d2 = cap^2 and argmin = its hint (a real admissible vertex, farther than
 * cap).  tuch_v2v_min_model_fix -- called with the SAME workspace once the inside test's flags `exterior` of THIS iteration
 * exist -- searches the columns again
 * that were cut off at the cap and are inside now (exhaustively: the exact result), and stores the flags as the next
 * prediction.  Together: exact (min_d2, argmin) for every vertex that is inside or has a partner within cap; cap must be
 * >= the loss's threshold (callers add 0.1 %: the loss recomputes the distance with its own rounding).
 * tuch_v2v_min_model_can_cap: does this model's search support it (leaf scan + hints; option v2v_cap)? */
int tuch_v2v_min_model_can_cap(const tuch_contact_model* model);
int tuch_v2v_min_model_capped(const tuch_contact_model* model, const float* verts, int B, float* min_d2, int32_t* argmin,
                              void* hint_inout, void* workspace, size_t workspace_bytes, int leave_room, void* zero,
                              size_t zero_bytes, const uint8_t* prev_exterior, float cap, void* stream);
int tuch_v2v_min_model_fix(const tuch_contact_model* model, int B, const uint8_t* exterior, uint8_t* prev_exterior, float cap,
                           float* min_d2, int32_t* argmin, void* hint_inout, void* workspace, size_t workspace_bytes,
                           void* stream);

/* Cluster tree over the faces of a closed mesh (host only, no device needed): the structure behind the
 * hierarchical evaluation of winding_numbers (tuch/utils/contact.py:112-147) inside tuch_exterior_flags.
 * A set of faces far from the query is replaced by a triangulation of its boundary loops, which subtends
 * the same solid angle for every query outside the set's bounding box (exact in real arithmetic).
 * nodes [num_nodes][8] = {cap_off, cap_len, exact_off, exact_len, skip, child0, child1, num_faces} in
 * preorder; vidx / sign [stream_len] = strip stream as in tuch_contact_model_strips (leaf strips in
 * [0, exact_len), then the caps); qperm [num_qblocks*128] = query order; frontier f =
 * frontier_nodes[frontier_off[f] .. frontier_off[f+1]) = subtrees that together cover the mesh;
 * launch_order [frontier_total * num_qblocks] = per frontier (at frontier_off[f] * num_qblocks) the
 * (subtree index << 16 | query block) pairs, the long-running ones first; rows [num_nodes][2] = (first
 * position, count) in qperm of the vertices below a node (each vertex belongs to one leaf); face_leaf [F] =
 * sequence number of the leaf holding each face (sort query points by it to make them coherent).
 * tuch_contact_model_create builds the same tree internally.  Fails (TUCH_ERR_ARG) for a mesh that is
 * not a closed, consistently oriented manifold; the library then keeps the flat evaluation. */
typedef struct tuch_cluster_tree tuch_cluster_tree;
int tuch_cluster_tree_build(int V, int F, const int32_t* faces, int leaf_faces, tuch_cluster_tree** out);
void tuch_cluster_tree_free(tuch_cluster_tree* tree);
int tuch_cluster_tree_info(const tuch_cluster_tree* tree, int* num_nodes, int* exact_len, int* stream_len,
                           int* num_qblocks, int* num_frontiers, int* frontier_total);
int tuch_cluster_tree_export(const tuch_cluster_tree* tree, int32_t* nodes, int32_t* vidx, float* sign,
                             int32_t* qperm, int32_t* frontier_off, int32_t* frontier_nodes,
                             int32_t* launch_order, int32_t* rows, int32_t* face_leaf);

/* exterior flags of losses.py:79-89 / loss.py:259-266: winding_numbers(verts, verts[faces]).le(thresh),
 * then BatchBodySegment.batch_has_self_isec (segmentation.py:117-124) and the re-marking of
 * vertices interior to their own segment.  verts [B,V,3] -> exterior [B,V] bytes;
 * optional: w [B,V], seg_w / seg_exterior [B, seg_q_total]. */
size_t tuch_exterior_workspace_bytes(const tuch_contact_model* model, int B);
int tuch_exterior_flags(const tuch_contact_model* model, const float* verts, int B, int apply_segments,
                        float thresh, float* w, uint8_t* exterior, float* seg_w, uint8_t* seg_exterior,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Winding numbers of arbitrary points against the model's mesh posed by `verts`, loss.py:295-297
 * (HD points offset along the face normals).  points [B,Q,3] (padded); counts [B] device ints or
 * NULL: meaningful points per body; padded entries give w = 0, exterior = 1. */
size_t tuch_winding_points_workspace_bytes(const tuch_contact_model* model, int B, int Q);
int tuch_winding_points(const tuch_contact_model* model, const float* verts, const float* points,
                        const int32_t* counts, int B, int Q, float thresh, float* w, uint8_t* exterior,
                        void* workspace, size_t workspace_bytes, void* stream);

/* region-pair minima: TUCH.contact_from_verts, train_module.py:69-91 (select NULL, unmasked)
 * and the region-to-region term, losses.py:107-117 (select = gt_contact & has_discrete_contact,
 * geodesically masked).  out_min [B,P] (0 where not selected), out_ij [B,P,2] arg-min vertices. */
int tuch_region_pair_min(const tuch_contact_model* model, const float* verts, int B, const uint8_t* select,
                         int use_geomask, float* out_min, int32_t* out_ij, void* stream);
int tuch_region_pair_min_bwd(const tuch_contact_model* model, const float* verts, int B, const int32_t* ij,
                             const float* grad_out, float* grad_verts, void* stream);

/* ---- SMPL forward / backward: tuch/models/smpl.py:34-56 over smplx 0.1.13 lbs() -------------
 * Model arrays are HOST pointers in the layouts smplx registers them: v_template [V,3],
 * shapedirs [V,3,10], posedirs [207,3V], J_regressor [24,V], lbs_weights [V,24], parents [24],
 * extra_vertex_ids [21] (smplx VertexJointSelector), J_regressor_extra [9,V] and joint_map [49]
 * (models/smpl.py:39-42). */
typedef struct tuch_smpl_model tuch_smpl_model;

int tuch_smpl_model_create(tuch_smpl_model** out, int V, const float* v_template, const float* shapedirs,
                           const float* posedirs, const float* J_regressor, const float* lbs_weights,
                           const int32_t* parents, const int32_t* extra_vertex_ids,
                           const float* J_regressor_extra, const int32_t* joint_map);
void tuch_smpl_model_destroy(tuch_smpl_model* model);
int tuch_smpl_model_info(const tuch_smpl_model* model, int* V);

/* SMPL.forward(betas, body_pose, global_orient, pose2rot): betas [B,10]; pose [B,72] axis-angle
 * (pose2rot != 0) or [B,24,3,3] rotation matrices -> vertices [B,V,3], joints [B,49,3].
 * The forward workspace keeps the intermediates tuch_smpl_backward needs. */
size_t tuch_smpl_forward_workspace_bytes(const tuch_smpl_model* model, int B);
int tuch_smpl_forward(const tuch_smpl_model* model, const float* betas, const float* pose, int pose2rot, int B,
                      float* verts, float* joints, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the pose as SMPL.forward's caller holds it (tuch/models/smpl.py:44-47: global_orient [B,3] / [B,1,3,3]
 * and body_pose [B,69] / [B,23,3,3], two tensors): no concatenated copy; row strides in floats, so views work too. */
int tuch_smpl_forward_split(const tuch_smpl_model* model, const float* betas, const float* global_orient,
                            int global_orient_stride, const float* body_pose, int body_pose_stride, int pose2rot, int B,
                            float* verts, float* joints, void* workspace, size_t workspace_bytes, void* stream);
/* Adjoint: g_verts [B,V,3] / g_joints [B,49,3] (either may be NULL) -> g_betas [B,10],
 * g_pose [B,72] or [B,24,3,3]. */
size_t tuch_smpl_backward_workspace_bytes(const tuch_smpl_model* model, int B);
int tuch_smpl_backward(const tuch_smpl_model* model, const float* pose, int pose2rot, int B,
                       const void* fwd_workspace, const float* g_verts, const float* g_joints, float* g_betas,
                       float* g_pose, void* workspace, size_t workspace_bytes, void* stream);
int tuch_smpl_backward_split(const tuch_smpl_model* model, const float* global_orient, int global_orient_stride,
                             const float* body_pose, int body_pose_stride, int pose2rot, int B,
                             const void* fwd_workspace, const float* g_verts, const float* g_joints, float* g_betas,
                             float* g_global_orient, int g_global_orient_stride, float* g_body_pose,
                             int g_body_pose_stride, void* workspace, size_t workspace_bytes, void* stream);
/* The same with a gradient the caller already holds for body_pose (g_body_pose_add: same shape, row stride in floats; or
 * NULL): g_body_pose = this call's gradient + that one.  In SMPLify-DC body_pose feeds the body model and the pose prior
 * (tuch/smplify/losses.py:63): autograd would add the two gradients in a separate launch. */
int tuch_smpl_backward_split_add(const tuch_smpl_model* model, const float* global_orient, int global_orient_stride,
                                 const float* body_pose, int body_pose_stride, int pose2rot, int B,
                                 const void* fwd_workspace, const float* g_verts, const float* g_joints, float* g_betas,
                                 float* g_global_orient, int g_global_orient_stride, float* g_body_pose,
                                 int g_body_pose_stride, const float* g_body_pose_add, int g_body_pose_add_stride,
                                 void* workspace, size_t workspace_bytes, void* stream, const void* g_verts_fixed);
/* g_verts_fixed (here and in tuch_smpl_backward_split_adam; or NULL): [B,V,3] 64-bit fixed-point sums (2^-36, the accumulators
 * tuch_smplify_stage2_fused leaves when it is called with grad_points = NULL in deterministic mode) ADDED to g_verts where the
 * skinning adjoint reads it -- the stage-2 tail's vertex gradient without a conversion launch on the step's serial tail. */
/* tuch_smpl_backward_split_add (axis-angle poses) + torch.optim.Adam's update (tuch_adam_step's arithmetic) of the two pose
 * tensors themselves, applied by the last backward kernel to the rows whose gradient it has just written: for a fit whose
 * optimiser holds exactly [global_orient, body_pose] and whose whole gradient arrives through this call (SMPLify-DC stage 2,
 * tuch/smplify/smplifydc.py:149-183: the body model + the pose prior via g_body_pose_add) -- the optimiser's own launch at
 * the end of every iteration is gone.  param_*: the tensors the optimiser updates (normally the memory global_orient /
 * body_pose point to), exp_avg* contiguous [B,3] / [B,69], step: the optimiser's device counter (advanced by one),
 * ticket: one zeroed int the call leaves zero.  The gradients are still written to g_*. */
int tuch_smpl_backward_split_adam(const tuch_smpl_model* model, const float* global_orient, int global_orient_stride,
                                  const float* body_pose, int body_pose_stride, int B, const void* fwd_workspace,
                                  const float* g_verts, const float* g_joints, float* g_betas, float* g_global_orient,
                                  int g_global_orient_stride, float* g_body_pose, int g_body_pose_stride,
                                  const float* g_body_pose_add, int g_body_pose_add_stride,
                                  float* param_global_orient, int param_global_orient_stride, float* param_body_pose,
                                  int param_body_pose_stride, float* exp_avg_global_orient, float* exp_avg_sq_global_orient,
                                  float* exp_avg_body_pose, float* exp_avg_sq_body_pose, float* step, int* ticket,
                                  float lr, float beta1, float beta2, float eps,
                                  void* workspace, size_t workspace_bytes, void* stream, const void* g_verts_fixed);

/* ---- caller-side glue of the training step ----------------------------------------
 * tuch_estimate_translation: tuch/utils/geometry.py:114-205 (estimate_translation + estimate_translation_np):
 * camera translation [B,3] that best projects the model joints onto the 2D keypoints (weighted least squares,
 * weights sqrt(conf)); joints3d [B,J,3], keypoints2d [B,J,3] = (u, v, conf), J = 49; has_anno[b] selects the
 * ground-truth joints [25,J) else the OpenPose joints [0,25); samples whose confidences sum to 0 get zeros.
 * float64 inside, like the reference's numpy.
 * tuch_rotmat_to_angle_axis: torchgeometry 0.1.2 rotation_matrix_to_angle_axis as called at
 * tuch/train/train_module.py:208-212 and demo_smplify_dc.py:128-132; rotmat [N,3,row_stride] with
 * row_stride 3 or 4 (the callers append a homogeneous column); NaN entries propagate and
 * are zeroed by the callers (train_module.py:212). */
int tuch_estimate_translation(const float* joints3d, const float* keypoints2d, const uint8_t* has_anno, int B, int J,
                              float focal_length, float img_size, float* trans, void* stream);
int tuch_rotmat_to_angle_axis(const float* rotmat, int N, int row_stride, float* angle_axis, void* stream);

/* ---- the HD-mesh branch of RegressorLoss.contact_loss, tuch/train/loss.py:274-301, as one device pipeline ----
 * (csrc/hd_contact.hip).  tuch_hd_model holds the HD vertex regressor (loss.py:81-83) as its three non-zeros per
 * row -- hd_idx / hd_w [N,3] -- and faces_vert_is_sampled_from (loss.py:85-87) -- hd_face [N]; all host arrays, copied.
 * It refers to the contact model (faces, geodesic mask, cluster tree), which must outlive it.  Point order is the
 * library's business (the loss is a sum over the points): tuch_hd_model_info reports it (order_host[k] = caller's
 * index of the k-th point). */
typedef struct tuch_hd_model tuch_hd_model;
int tuch_hd_model_create(tuch_hd_model** out, const tuch_contact_model* contact_model, int N, const int32_t* hd_idx,
                         const float* hd_w, const int32_t* hd_face);
/* The same for a regressor with up to K (1..8) non-zeros per row: hd_idx / hd_w [N,K], rows with fewer non-zeros padded
 * with weight 0 (any valid vertex id).  The reference multiplies the DENSE matrix (loss.py:285): a regressor file that is
 * not a plain barycentric sampling (<= 3 non-zeros) still loads. */
int tuch_hd_model_create_k(tuch_hd_model** out, const tuch_contact_model* contact_model, int N, int K,
                           const int32_t* hd_idx, const float* hd_w, const int32_t* hd_face);
void tuch_hd_model_destroy(tuch_hd_model* model);
int tuch_hd_model_info(const tuch_hd_model* model, int* N, int32_t* order_host);
/* One call = loss.py:274-315 for the whole batch: exterior [B,V] u8, min_d2 [B,V], partner [B,V] int32 are the
 * vertex-level results for the same verts [B,V,3] (tuch_exterior_flags with the segment filter, tuch_v2v_min_model);
 * valid [B] u8 or NULL.  terms [B,2] = (sum over interior HD points of tanh^2(d/0.04), sum over exterior ones of
 * 0.005 tanh^2(d/0.005)); bodies with valid == 0 or without a selected point give (0, 0).  thresh = 0.99.
 * `saved` (tuch_hd_contact_saved_bytes, caller-owned, opaque) carries the selection to tuch_hd_contact_bwd, which
 * OVERWRITES grad_verts [B,V,3] with d(sum_b grad_terms[b,:] . terms[b,:]) / d verts.  All buffers are sized for the
 * worst case (every HD point selected); the actual counts never leave the device: no synchronisation, no allocation. */
size_t tuch_hd_contact_saved_bytes(const tuch_hd_model* model, int B);
size_t tuch_hd_contact_workspace_bytes(const tuch_hd_model* model, int B);
int tuch_hd_contact_fwd(const tuch_hd_model* model, const float* verts, const uint8_t* exterior, const float* min_d2,
                        const int32_t* partner, const uint8_t* valid, int B, float euclthres, float thresh, float* terms,
                        void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes, void* stream);
int tuch_hd_contact_bwd(const tuch_hd_model* model, const void* saved, const float* grad_terms, int B,
                        float* grad_verts, void* workspace, size_t workspace_bytes, void* stream);
/* inspection (tests; synchronous copies): counts_host [B]; selected_host [B,N] = caller-order index of the point in
 * every slot, -1 beyond the count (or NULL) */
int tuch_hd_contact_selection(const tuch_hd_model* model, const void* saved, int B, int32_t* counts_host,
                              int32_t* selected_host);
/* per slot (order of tuch_hd_contact_selection): caller-order index of the partner the search found (-1 beyond the
 * count) and the exterior flag of the point; both [B,N] */
int tuch_hd_contact_details(const tuch_hd_model* model, const void* saved, int B, int32_t* partner_host, uint8_t* ext_host);

/* HD points of tuch/train/loss.py:285 (hd = Vert_Regressor[selected] @ verts, a dense [N_hd,6890] matrix with three
 * non-zeros per row): point n belongs to body body_of_point[n] and is HD point hd_of_point[n];
 * points[n] = sum_k hd_w[h][k] * verts[body][hd_idx[h][k]].  verts [B,V,3], hd_idx / hd_w [N_hd,3], points [N,3].
 * The adjoint ADDS into grad_verts [B,V,3] (float atomics; zero it first). */
int tuch_hd_points_fwd(const float* verts, const int32_t* body_of_point, const int32_t* hd_of_point,
                       const int32_t* hd_idx, const float* hd_w, int V, int N, float* points, void* stream);
int tuch_hd_points_bwd(const float* grad_points, const int32_t* body_of_point, const int32_t* hd_of_point,
                       const int32_t* hd_idx, const float* hd_w, int V, int N, float* grad_verts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TUCH_AMD_H */
