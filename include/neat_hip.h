/* neat_hip.h -- C ABI of libneat_hip.so: the MI355X (gfx950) implementation of NEAT's volumetric-
 * rendering hot path.  Plain pointers and sizes only (no torch types): every pointer is a DEVICE
 * pointer to float32 unless stated, `stream` is a hipStream_t passed as void*, every call is
 * asynchronous on that stream, never synchronises, never allocates, never keeps caller memory.
 * Return value: 0 on success, otherwise a hipError_t (or -1 for bad arguments).
 *
 * The reference (cherubicXN/neat) has no native layer: these entry points replace the ATen op
 * sequences issued by the Python lines cited on each function (paths relative to the reference's
 * code/ directory).  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Layouts: "row-major [P,C]" is the torch layout the reference uses.  Workspaces are opaque float
 * arenas sized by the *_ws_floats() queries; they carry the activations saved for backward.
 */
#ifndef NEAT_HIP_H
#define NEAT_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NEAT_NUM_LAYERS 19   /* implicit_network.lin0..8, rendering_network.lin0..4, attraction_network.lin0..4 */
#define NEAT_FEATURE 256

/* Raw parameters exactly as in the reference state_dict (weight-normed Linear layers:
 * model/networks/neat_wfr_rend_a.py:71-72,167-168,227-228): weight_v [out,in], weight_g [out,1], bias [out]. */
typedef struct neat_net_params {
  const float* v[NEAT_NUM_LAYERS];
  const float* g[NEAT_NUM_LAYERS];
  const float* b[NEAT_NUM_LAYERS];
} neat_net_params;

/* Gradient destinations, same shapes as neat_net_params (overwritten, not accumulated).  A NULL dv[l]
 * skips layer l (e.g. the heads when only the SDF network was run). */
typedef struct neat_net_grads {
  float* dv[NEAT_NUM_LAYERS];
  float* dg[NEAT_NUM_LAYERS];
  float* db[NEAT_NUM_LAYERS];
} neat_net_grads;

int neat_abi_version(void);      /* 13 */

/* `precision` selects the build of the GEMM-class kernels:
 *   NEAT_F32  (0): exact-f32 MFMA, fp32 activations  -- parity build (outputs within 1e-4 of the reference)
 *   NEAT_BF16 (1): bf16 MFMA with fp32 accumulate, bf16 hidden activations -- throughput build
 *   NEAT_BF16X3 (2): the NEAT_F32 build (same layouts, workspaces, packed weights, kernels) whose two GEMM kernels evaluate every
 *                    product as three bf16 MFMAs on hi/lo splits of both operands, fp32 accumulate (~2^-17 relative per product):
 *                    fp32-grade parity at a multiple of the f32-MFMA rate
 *   NEAT_F16 (3): the NEAT_BF16 build with IEEE half instead of bf16 as the 16-bit type: v_mfma_f32_32x32x16_f16, fp32 accumulate,
 *                 f16 hidden activations (BASELINE config 5, "fp16 MFMA with fp32 accumulate").  Same kernels, layouts, workspaces
 *                 and speed; 3 more mantissa bits (errors ~8x below NEAT_BF16's), 5 exponent bits: the backward pass runs on
 *                 cotangents scaled by 4096 internally (scaled where the caller's cotangents are read, scaled back on the finished
 *                 gradients; nothing visible at this interface).
 *   NEAT_F16X3 (4): the NEAT_F16 build (layouts, workspaces + lo planes, the whole backward pass) whose three FORWARD chains -- SDF
 *                   primal (rend_a :78-96), SDF adjoint = normals (:121-127), both heads (:139-255) -- are fused launches that
 *                   evaluate every product as three f16 MFMAs on hi/lo splits of both operands (~22 mantissa bits): forward outputs
 *                   within 1e-4 of the reference like NEAT_F32, gradients at the fp32 build's bar (the backward pass reads the hi
 *                   planes = exactly what NEAT_F16 saves), at about the speed of NEAT_F16.
 *   NEAT_F16X3_FASTVALUES (5): accepted by the VALUES-mode calls only (neat_sdf_forward mode 0, neat_sdf_values_gated,
 *                   neat_sdf_ws_floats) together with a NEAT_F16X3 pack: the query runs through NEAT_F16's one-product chain (3x
 *                   faster).  For a depth sampler that may trade the reference's exact samples for speed (the sampled distribution
 *                   stays the reference's to ~1e-4 of the depth range, tests/test_gpu_parity.py::test_sampler_vs_reference_golden).
 * Packed weights, workspaces and forward/backward calls of one pass must use the same value. */
#define NEAT_F32 0
#define NEAT_BF16 1
#define NEAT_BF16X3 2
#define NEAT_F16 3
#define NEAT_F16X3 4
#define NEAT_F16X3_FASTVALUES 5

/* ---- a15: weight norm + packing (replaces the per-call `_weight_norm` pre-hook) -------------------
 * Computes W = g * v/|v| for all 19 layers once per step and stores W and W^T in MFMA-fragment order. */
size_t neat_packed_floats(int precision);
int neat_pack_weights(const neat_net_params* net, float* packed, int precision, void* stream);

/* ---- a1: pixel -> unit ray directions (utils/rend_util.py:55-81 get_camera_params, :95-108 lift) --
 * uv [R,2], pose [4,4] cam-to-world, K row stride `kstride` (3 or 4).  dirs [R,3].  Origin = pose[:3,3]. */
int neat_camera_rays(const float* uv, const float* pose, const float* K, int kstride, int R, float* dirs, float* origins /* [R,3] or NULL: the camera centre per ray */,
                     void* stream);
/* a12: the eikonal points of a training step (neat_wfr_rend_a.py:515-527) as one array [2R + J, 3]:
 * [uniform [R,3] (drawn by the caller) | origins + z_eik dirs | extra [J,3] (the global junctions, J may be 0)].
 * z_eik [R] = the depth drawn per ray; NULL (ABI v12): picked here as z[r, idx[r]] from the ray's S depths z [R,S] with the drawn
 * indices idx [R] (int64) -- the `torch.gather(z_vals, 1, idx)` of model/ray_sampler.py:276-277 without a launch of its own. */
int neat_eik_points(const float* uniform, const float* origins, const float* dirs, const float* z_eik, const float* extra, int R, int J,
                    float* out, const float* z, int S, const long long* idx, void* stream);

/* ---- a4+a5: SDF / implicit network (neat_wfr_rend_a.py:78-137) -----------------------------------
 * mode 0: values only  -> sdf[P] = get_sdf_vals(x)            (:131-137), nothing saved
 * mode 1: outputs      -> forward() [P,257], clamped sdf, feature, d sdf/dx (get_outputs :111-129 /
 *                         gradient :98-109 when radius <= 0), activations saved in `ws` for backward.
 * radius > 0 enables the bounding-sphere clamp min(sdf, scale*(radius-|x|)).
 * Any of out257 / sdf / feat / grad may be NULL.  x is row-major [P,3]. */
size_t neat_sdf_ws_floats(int P, int mode, int precision);
int neat_sdf_forward(const float* packed, const neat_net_params* net, const float* x, int P, int mode, int precision,
                     float radius, float scale, float* ws,
                     float* out257, float* sdf, float* feat, float* grad, void* stream);
/* Backward of mode-1 forward (autograd incl. the double backward through d sdf/dx, which the reference
 * gets from create_graph=True at :121-127).  Cotangents are row-major and may be NULL (= zero):
 * d_out257 [P,257] (of forward()), d_sdf [P] (of the clamped sdf), d_feat [P,256], d_grad [P,3].
 * Writes grads->dv/dg/db[0..8]. */
int neat_sdf_backward(const float* packed, const neat_net_params* net, float* ws, int P, int precision,
                      const float* d_out257, const float* d_sdf, const float* d_feat, const float* d_grad,
                      const neat_net_grads* grads, void* stream);

/* ---- a6+a7: the two heads on given inputs (RenderingNetwork.forward :235-255,
 * AttractionFieldNetwork.forward :175-197), row-major inputs; rgb [P,3], lines [P,2,3]. */
size_t neat_heads_ws_floats(int P, int precision);
int neat_heads_forward(const float* packed, const neat_net_params* net, const float* points, const float* normals,
                       const float* view_dirs, const float* feats, int P, int precision, float* ws,
                       float* rgb, float* lines, void* stream);

/* ---- a5-a10 fused main pass: VolSDFNetwork.forward :392-422 (+ :530-536 normal_map in eval) -------
 * origins/dirs [R,3], z [R,S] sorted depths, beta = DEVICE pointer to one float (a pointer so that no host sync is needed); the
 * density's beta is |*beta| + beta_min (model/density.py:28-30; ABI v12): pass the raw parameter density.beta and the module's
 * beta_min, or -- as up to v11 -- the value of get_beta() and 0.
 * Outputs (row-major, NULL to skip where noted): points [R,S,3] (opt), weights [R,S] (opt),
 * sdf [R,S] (opt), rgb [R,3], lines3d [R,2,3], depth [R], xyz [R,3], normal_map [R,3] (opt). */
size_t neat_render_ws_floats(int R, int S, int E, int precision);
int neat_render_forward(const float* packed, const neat_net_params* net, const float* origins, const float* dirs,
                        const float* z, int R, int S, int precision, const float* beta, float beta_min, float radius, float scale, float* ws,
                        float* points, float* weights, float* sdf, float* rgb, float* lines3d, float* depth,
                        float* xyz, float* normal_map, const float* eik_points, int E, float* eik_grad, void* stream);
/* a12 folded in: `eik_points` [E,3] (may be NULL with E = 0) are E extra points appended to the R*S ray samples for
 * the SDF network only; eik_grad [E,3] receives ImplicitNetwork.gradient (:98-109, no sphere clamp) at those points.
 * This removes ~66 latency-bound small launches per training step (the 2R eikonal points of :515-527).
 *
 * Backward of neat_render_forward.  Cotangents d_rgb [R,3], d_lines3d [R,6], d_depth [R], d_xyz [R,3], d_eik_grad [E,3],
 * d_acc [R] = cotangent of the ray's opacity acc_map = sum_i w_i (the white_bkgd term of :411-413; ABI v11) (NULL = zero).  lines3d uses detached weights exactly as :410.  Writes all 19 layers' grads and the
 * per-ray partial derivative wrt the density's beta, dbeta_ray [R].  dbeta (device pointer to one float, or NULL; ABI v12) receives
 * the gradient of *beta itself: sgn(*beta) * sum_r dbeta_ray[r], summed in a fixed order by one more tiny launch (the `.sum()` and the
 * backward of `.abs()` of density.py:29-30). */
int neat_render_backward(const float* packed, const neat_net_params* net, float* ws, const float* dirs,
                         const float* z, int R, int S, int E, int precision, const float* beta, float beta_min,
                         const float* d_rgb, const float* d_lines3d, const float* d_depth, const float* d_xyz,
                         const float* d_eik_grad, const float* d_acc, const neat_net_grads* grads, float* dbeta_ray, float* dbeta, void* stream);

/* Forward-only variant for eval / inference callers (code/neat-final-parsing.py:203-218 drives `model(s)` in 2048-ray chunks under
 * model.eval(); training/volsdf_train.py:312-320 renders validation images the same way): same arguments and results as
 * neat_render_forward with E = 0, but the workspace keeps nothing for a backward pass (hidden activations of the adjoint chain and
 * of the two heads ping-pong between two buffers): about a quarter of neat_render_ws_floats. */
size_t neat_render_eval_ws_floats(int R, int S, int precision);
int neat_render_forward_eval(const float* packed, const neat_net_params* net, const float* origins, const float* dirs,
                             const float* z, int R, int S, int precision, const float* beta, float beta_min, float radius, float scale, float* ws,
                             float* points, float* weights, float* sdf, float* rgb, float* lines3d, float* depth,
                             float* xyz, float* normal_map, void* stream);

/* ---- a2: UniformSampler.get_z_vals (model/ray_sampler.py:61-95): N depths per ray between near and far (per ray [R] if the pointer
 * is given, else the scalar), stratified jitter with rnd [R,N] in training (NULL: the plain grid).  t [N] = the linspace(0, 1, N)
 * grid.  Operations rounded one by one like the reference's elementwise ops: bit-identical to them. */
int neat_uniform_depths(const float* near_r, float near_s, const float* far_r, float far_s, const float* t, const float* rnd, int R, int N,
                        float* z, void* stream);

/* ---- a13: sample_pdf (model/ray_sampler.py:16-59) and the sort of UniformSampler.get_z_vals_fine (:97-106), one wavefront per ray:
 * bins [R,nb], weights [R,nb-1] (the caller passes weights[..., 1:-1] and the interval mid points), u [N] (u_stride 0) or [R,N]
 * -> samples [R,N]; z_merge [R,nz] given: z_out [R,nz+N] = sort(cat[z_merge, samples]).  fp64 CDF rounded once per knot. */
int neat_sample_pdf(const float* bins, const float* weights, int nb, int R, const float* u, int u_stride, int N, float* samples,
                    const float* z_merge, int nz, float* z_out, void* stream);

/* ---- a3: ErrorBoundSampler bookkeeping (model/ray_sampler.py:130-293), one wavefront per ray --------------------
 * One round of Algorithm 1 = neat_sdf_forward(mode 0) on the new samples, then:
 *  neat_sampler_bound    : merge the sdf values (order from the previous round, :152-157), d* per interval (:161-173),
 *                          per-ray bisection of beta on [beta0, beta_in] (:177-185, get_error_bound :285-293);
 *                          *flag |= 1 if any ray still has beta > beta0 (the batch-global test of :200 -- the host
 *                          reads the flag once per round, as the reference syncs once per round).
 *  neat_sampler_resample : pdf (error-bound opacity :205-215 if refine, rendering weights + 1e-5 :217-226 otherwise),
 *                          CDF, inverse-CDF sampling at u (:237-249); refine also emits the sorted union with the old
 *                          grid and its gather order (:254).  u has N entries per ray (u_stride = N) or is shared (0).
 *  neat_sampler_finish   : [final samples | near | far | picked grid points] sorted (:259-272) and the eikonal depth.
 * All tensors row-major; n, N <= 1024. */
int neat_sampler_bound(const float* z, int n, int R, const float* sdf_old, const float* sdf_new, const int* order, int n_old,
                       const float* beta_in, const float* beta0, float eps, int iters, float* sdf_out, float* beta_out,
                       int* flag, void* stream);
int neat_sampler_resample(const float* z, const float* sdf, int n, int R, const float* beta, int refine, float add_tiny,
                          const float* u, int u_stride, int N, float* samples, float* z_merged, int* order, void* stream);
int neat_sampler_finish(const float* samples, int N, const float* z, int n, const int* pick, int n_extra, float near, float far,
                        int R, const int* eik_idx, float* z_vals, float* z_eik, void* stream);

/* ---- a3 without host synchronisation (HIP-graph capturable) -----------------------------------------------------
 * The reference reads `beta.max() > beta0` on the host once per round (:200).  Here the rounds are a FIXED sequence of max_rounds
 * launches and the decision lives in device memory: int32 open[max_rounds] (zero-initialised; the bound kernel of round k ORs into
 * open[k]) and int32 cont[max_rounds] (the resample launch of round k writes 1 = "refined, go on" or 2 = "final samples drawn").
 * A launch of round k > 0 does nothing unless cont[k-1] == 1, so the rounds after the final one cost only their launch.
 *  neat_sdf_values_gated     : neat_sdf_forward(mode 0) that returns at once unless *gate == gate_value (gate = &cont[k-1], 1).
 *  neat_sdf_values_rays      : the same for the points o + z d of R rays x S depths (sdf [R S]): the round's `cam_loc + samples * dirs`
 *                              (:146) is formed by the launch that lays the points out for the SDF kernels (ABI v12).
 *  neat_sampler_init         : what precedes the first round (:131-143) in one launch (ABI v12): beta0[0] = |*beta| + beta_min
 *                              (density.py:29-30), beta_ray[r] = sqrt(beta_c sum_i (z[r,i+1] - z[r,i])^2) with beta_c = 1 / (4 log(1 + eps)),
 *                              and the nctl control words of the rounds zeroed.
 *  neat_sampler_bound_dev    : neat_sampler_bound with `open` = &open[k] and the same gate.
 *  neat_sampler_resample_dev : decides refine = *open && round + 1 < max_rounds on the device.  refine: N_refine samples at u_refine
 *                              (shared by the rays), merged grid and order, as neat_sampler_resample(refine=1).  Otherwise the N_final
 *                              output samples at u_final (stride u_final_stride), a copy of this round's grid into z_final
 *                              [R, ld_final] and its size into *n_final.
 *  neat_sampler_finish_dev   : picks the n_extra grid points on the device (sizes come from *n_final): keys == NULL ->
 *                              linspace(0, n-1, n_extra).long() (:266, eval); keys [>= n] -> the indices of the n_extra smallest
 *                              keys, a uniformly random subset like randperm(n)[:n_extra] (:264, training); then as
 *                              neat_sampler_finish.  `pick` [n_extra] receives the indices. */
int neat_sdf_values_gated(const float* packed, const neat_net_params* net, const float* x, int P, int precision, float radius,
                          float scale, float* ws, float* sdf, const int* gate, int gate_value, void* stream);
int neat_sdf_values_rays(const float* packed, const neat_net_params* net, const float* origins, const float* dirs, const float* z, int R,
                         int S, int precision, float radius, float scale, float* ws, float* sdf, const int* gate, int gate_value,
                         void* stream);
int neat_sampler_init(const float* z, int R, int n, const float* beta, float beta_min, float beta_c, float* beta0, float* beta_ray,
                      int* ctl, int nctl, void* stream);
int neat_sampler_bound_dev(const float* z, int n, int R, const float* sdf_old, const float* sdf_new, const int* order, int n_old,
                           const float* beta_in, const float* beta0, float eps, int iters, float* sdf_out, float* beta_out,
                           int* open, const int* gate, int gate_value, void* stream);
int neat_sampler_resample_dev(const float* z, const float* sdf, int n, int R, const float* beta, float add_tiny,
                              const float* u_refine, int N_refine, float* samples_refine, float* z_merged, int* order,
                              const float* u_final, int u_final_stride, int N_final, float* samples_final, float* z_final, int ld_final,
                              int* n_final, const int* open, int* cont, int round, int max_rounds, void* stream);
int neat_sampler_finish_dev(const float* samples, int N, const float* z_final, int ld_final, const int* n_final, const float* keys,
                            int n_extra, int* pick, float near, float far, int R, const int* eik_idx, float* z_vals, float* z_eik,
                            void* stream);

/* ---- a3, one launch per round (ABI v13; replaces bound_dev + resample_dev + the layout launch of neat_sdf_values_rays per round) ------
 * The same arithmetic as the entry points above, reference model/ray_sampler.py:145-254 per round.  Control words
 * ctl = int32 open[max_rounds] | ran[max_rounds] | n_final (zeroed by neat_sampler_init_rays): round k ORs 1 into open[k] when a ray's
 * beta is still above beta0 (the batch-global test of :200); a launch of round k > 0 (this one and the round's SDF query, gate =
 * &open[k-1], gate_value 1) does nothing unless open[k-1] is set.  Because the test cannot be read inside the launch that produces it,
 * a round prepares both outcomes: every ray refines (rounds before the last), and the rays whose own beta has reached beta0 -- all
 * rays in the last round -- also draw the final samples, copy the grid to z_final and write n_final; the last round that ran wins.
 *  neat_sdf_ldp             : point stride of the SDF kernels' feature-major layout for P points (the first 3 rows of a
 *                             neat_sdf_ws_floats workspace are x [3][ldp]).
 *  neat_sdf_values_laid_out : neat_sdf_values_gated on points that already sit in the workspace's x rows.
 *  neat_sampler_init_rays   : neat_sampler_init + (x_fm != NULL) the first round's query points cam_loc + z dirs (:146) into x_fm,
 *                             + (keys != NULL) for every possible final grid size n_step (c + 1), c < n_cand, the n_extra smallest
 *                             of its first n keys in key order -> pick_all [n_cand][n_extra] (the training-mode randperm(n)[:n_extra]
 *                             of :264-265 as neat_sampler_finish_dev draws it, computed beside the prologue instead of behind the rounds).
 *  neat_sampler_round       : merge + d* + bisection (as neat_sampler_bound), then refine: N_refine samples at u_refine, merged grid
 *                             z_merged / order_out [R, n + N_refine], and the NEXT round's query points into x_fm [3][ldp] (the
 *                             workspace of the next neat_sdf_values_laid_out; NULL: not wanted); final: N_final samples at u_final.
 *  neat_sampler_finish_picked : neat_sampler_finish with the grid size read from *n_final and the picks from
 *                             pick_all[n / n_step - 1] (pick_all == NULL: linspace(0, n - 1, n_extra).long(), eval mode :266). */
int neat_sdf_ldp(int P, int precision);
int neat_sdf_values_laid_out(const float* packed, const neat_net_params* net, int P, int precision, float radius, float scale, float* ws,
                             float* sdf, const int* gate, int gate_value, void* stream);
int neat_sampler_init_rays(const float* z, int R, int n, const float* beta, float beta_min, float beta_c, float* beta0, float* beta_ray,
                           int* ctl, int nctl, const float* origins, const float* dirs, float* x_fm, int ldp, const float* keys, int n_step,
                           int n_cand, int n_extra, int* pick_all, void* stream);
int neat_sampler_round(const float* z, int n, int R, const float* sdf_old, const float* sdf_new, const int* order, int n_old,
                       const float* beta_in, const float* beta0, float eps, int iters, float* sdf_out, float* beta_out, int* ctl, int round,
                       int max_rounds, float add_tiny, const float* u_refine, int N_refine, float* samples_refine, float* z_merged,
                       int* order_out, const float* origins, const float* dirs, float* x_fm, int ldp, const float* u_final,
                       int u_final_stride, int N_final, float* samples_final, float* z_final, int ld_final, void* stream);
int neat_sampler_finish_picked(const float* samples, int N, const float* z_final, int ld_final, const int* n_final, const int* pick_all,
                               int n_step, int n_extra, float near, float far, int R, const int* eik_idx, float* z_vals, float* z_eik,
                               void* stream);

/* ---- 8f-1 (next row): dataset attraction field, replacement for the un-vendored hawp.base._C.encodels ------------
 * (datasets/blender_hawp_dataset.py:96, scene_hawp_dataset.py:95).  lines [N,4] = (x1,y1,x2,y2) in pixels;
 * lmap [6,H,W] = closest point - pixel (0:2), endpoint 1 - pixel (2:4), endpoint 2 - pixel (4:6), all (x,y);
 * label [H,W] int32 = index of the nearest segment (the reference's labels_onehot.max(dim=0)[1]);
 * valid [H,W] uint8 (may be null) = the reference's labels_onehot.max(dim=0)[0], which the dataset multiplies into its support mask
 *   (blender_hawp_dataset.py:98,130): 1 where the pixel HAS a nearest segment -- i.e. at least one segment with finite coordinates
 *   exists; 0 everywhere for N = 0 (then lmap = 0, label = 0) and for segments that are all non-finite.  A zero-length segment is a
 *   segment (a point).
 * PARITY UNPINNED: hawp is an empty submodule here; semantics are those the call sites rely on. */
int neat_encode_lines(const float* lines, int N, int H, int W, float* lmap, int* label, unsigned char* valid, void* stream);

/* ---- 8f-1, batch assembly on the device: the ray sampling of Dataset.__getitem__ (datasets/blender_hawp_dataset.py:186-198,
 * scene_hawp_dataset.py:179-190) with the view's maps resident in HBM.  pool [npool] int32 = pixels of the line support
 * (mask.nonzero()); draw [n] int64 = the step's draws into the pool (made on the host: np.random.choice's stream for the ABC class);
 * att [HW,2] foot points, rgb [HW,3], labels [HW] int32 (nearest segment), lines [nlines,5].  Per ray i, pixel p = pool[draw[i]]:
 * uv [n,2] = (p mod W, p div W), uv_proj [n,2] = att[p], rgb_out [n,3], lines_out [n,5] = lines[labels[p]], labels_out [n],
 * pixel_out [n] = p. */
int neat_gather_batch(const int* pool, int npool, const long long* draw, int n, int W, const float* att, const float* rgb, const int* labels,
                      const float* lines, int nlines, float* uv, float* uv_proj, float* rgb_out, float* lines_out, long long* labels_out,
                      long long* pixel_out, void* stream);

/* ---- step prefix: the fresh batch's tensors (datasets' collate output, training/volsdf_train.py:361-364 `model_input[...] = ...cuda()`)
 * and the step's CPU-drawn randoms (the reference draws on the host: model/ray_sampler.py:86-93,132; neat_wfr_rend_a.py:330-335) into the
 * persistent tensors a captured step reads, as ONE launch: copy i moves nbytes[i] bytes from src[i] to dst[i] (n <= 16).  A source may be
 * device memory or pinned host memory (hipHostMalloc: the kernel reads it over the bus -- no separate copy-engine transfer and none of
 * the idle gaps between a copy and the next kernel).  Pointers 4-byte aligned, sizes multiples of 4. */
int neat_copy_batch(const void* const* src, void* const* dst, const long long* nbytes, int n, void* stream);

/* ---- a11 / a14 glue as single launches.  neat_project2d = VolSDFNetwork.project2D (model/networks/neat_wfr_rend_a.py:317-326):
 * K [3,3] and w2c [3,4] = [R|T] row-major on the device, X [N,3] -> uv [N,2]; its backward gives d_X from d_uv.
 * neat_line_loss = VolSDFLoss.get_line_loss (model/networks/loss_wfr.py:34-45): pred, gt [R,4], weight [R] ->
 * out2 = {loss, number of gated lines}, per_line [R], d_pred [R,4] = d loss / d pred. */
/* junction block pieces (model/networks/neat_wfr_rend_a.py:441-489) as single launches: neat_l3d (:441-447: plane intersection per ray),
 * neat_junction_cost (:472: cost[v][c] = |cand2d[c] - gt2d[v]|), neat_junction_gate (:474-489: matched costs of the neat_lsap pairs,
 * median or 10 px gate, matched candidates gathered into padded [K,.] arrays + mask; K <= 2048). */
int neat_l3d(const float* x, const float* o, const float* d, const float* normal, int R, float* l3d, void* stream);
int neat_junction_cost(const float* cand2d, const float* gt2d, int V, int C, float* cost, void* stream);
int neat_junction_gate(const long long* rows, const long long* cols, int K, const float* cost, int C, const float* cand3d,
                       const float* cand2d, const float* cand2d_calib, int use_median, float* median, unsigned char* good, float* j3d,
                       float* j2d, float* j2d_calib, void* stream);
/* the tail of VolSDFLoss.forward (model/networks/loss_wfr.py:68-137) as two launches around the junction matching (neat_lsap):
 * neat_loss_terms -> scal[0] = L1 rgb loss (mean), scal[1] = eikonal loss, d_rgb [R,3], d_gtheta [E,3] (cotangents for unit
 * upstream gradient), pair_cost [K,J] = cdist_1(loc3, glo3) + 0.1 cdist_1(loc2c, glo2c);
 * neat_loss_pairs (ri, ci, n_match from neat_lsap over pair_cost) -> scal[2..5] = mean 3-D / calibrated 2-D / pixel 2-D pair
 * distance and the count of pairs with cost < 10, scal[6] = rgb + w_eik eik + w_line line_loss[0] + w_j3 j3d + w_j2 j2d;
 * d_glo3 [J,3], d_glo2c [J,2] = cotangents of the global junctions.
 * ABI v12: the cotangents can leave as gradients of the TOTAL loss scal[6], so that the backward pass has nothing left to multiply:
 * d_gtheta carries eik_grad_scale (pass w_eik, or 1), d_glo3 / d_glo2c carry w_j3 / w_j2 when weighted_grads != 0; `total` (or NULL)
 * receives scal[6] once more, in an allocation of its own (the differentiable output of the caller's autograd node). */
int neat_loss_terms(const float* rgb, const float* rgb_gt, int R, const float* gtheta, int E, const float* loc3, const float* loc2c, int K,
                    const float* glo3, const float* glo2c, int J, float* scal, float* d_rgb, float* d_gtheta, float* pair_cost,
                    float eik_grad_scale, void* stream);
/* ABI v13: neat_line_losses and neat_loss_terms (independent of each other) as the two workgroups of one launch; same arithmetic.
 * d_lines3d != NULL: pred_calib = neat_project2d(identity, w2c, lines3d [L,2,3]) (rend_a :441); the gradient of the line term is carried
 * through that projection here (d_lines3d [L,2,3], neat_project2d_backward's arithmetic).  Likewise neat_loss_pairs with w2c != NULL:
 * glo2c = neat_project2d(identity, w2c, glo3) (:496) and d_glo3 receives d_glo2c's share -- the projections' backward launches and the
 * accumulation of the global junctions' two gradients disappear from the step. */
int neat_loss_lines_terms(const float* pred_px, const float* pred_calib, const float* gt5, const float* Kmat, int L, float threshold, float* out3,
                          float* d_pred_calib, float grad_scale, const float* rgb, const float* rgb_gt, int R, const float* gtheta, int E,
                          const float* loc3, const float* loc2c, int K, const float* glo3, const float* glo2c, int J, float* scal, float* d_rgb,
                          float* d_gtheta, float* pair_cost, float eik_grad_scale, const float* w2c, const float* lines3d, float* d_lines3d,
                          void* stream);
int neat_loss_pairs(const long long* ri, const long long* ci, const int* n_match, int Kmax, const float* loc3, const float* loc2c,
                    const float* loc2, const float* glo3, const float* glo2c, const float* glo2, int J, const float* pair_cost, float* scal,
                    float* d_glo3, float* d_glo2c, const float* line_loss, float w_eik, float w_line, float w_j3, float w_j2, int weighted_grads,
                    float* total, const float* w2c, void* stream);
/* global-junction MLP ffn(latents) (rend_a :303-313, :491): x [J,256] -> relu(W0 x + b0) -> relu(W1 . + b1) -> W2 . + b2 = y [J,3];
 * torch nn.Linear layouts (W [out,in]); h1, h2 [J,256] are saved for the backward; ws2 = 2 J 256 floats of scratch. */
int neat_ffn_forward(const float* x, int J, const float* W0, const float* b0, const float* W1, const float* b1, const float* W2,
                     const float* b2, float* h1, float* h2, float* y, void* stream);
int neat_ffn_backward(const float* x, int J, const float* W0, const float* W1, const float* W2, const float* h1, const float* h2,
                      const float* dy, float* ws2, float* dx, float* dW0, float* db0, float* dW1, float* db1, float* dW2, float* db2,
                      void* stream);
/* inverse of one n x n matrix (n <= 4, row stride lda): pose.inverse() (rend_a :440), K.inverse() (loss_wfr.py:59) */
int neat_inv_small(const float* A, int n, int lda, float* out, void* stream);
/* what the junction block's projections read (rend_a :424-431, :440) in one launch (ABI v12): w2c [3,4] = the first three rows of
 * pose^-1 (pose [4,4] cam-to-world, contiguous) and K3 [3,3] = the contiguous copy of the intrinsics' 3x3 block (row stride kstride) */
int neat_camera_mats(const float* pose, const float* K, int kstride, float* w2c, float* K3, void* stream);
int neat_project2d(const float* K, const float* w2c, const float* X, int N, float* uv, void* stream);
int neat_project2d_backward(const float* K, const float* w2c, const float* X, int N, const float* d_uv, float* d_X, void* stream);
/* ABI v13: the junction block's camera-only work and its double projections as single launches.
 *  neat_camera_setup   : neat_camera_rays for `uv` (dirs, origins) and -- uv2 != NULL -- for `uv2` (dirs2; rend_a :444), plus
 *                        neat_camera_mats (w2c [3,4], K3 [3,3]); same arithmetic as the three launches.
 *  neat_project2d_pair : neat_project2d of the same points with K (-> uv) and with K2 (-> uv2): rend_a :436-441 (lines2d / lines2d_calib),
 *                        :469-471 (junction candidates), :493-496 (global junctions) project with the intrinsics and with the identity. */
int neat_camera_setup(const float* uv, const float* uv2, const float* pose, const float* K, int kstride, int R, float* dirs, float* origins,
                      float* dirs2, float* w2c, float* K3, void* stream);
int neat_project2d_pair(const float* K, const float* K2, const float* w2c, const float* X, int N, float* uv, float* uv2, void* stream);
int neat_line_loss(const float* pred, const float* gt, const float* weight, int R, float threshold, float* out2, float* per_line,
                   float* d_pred, void* stream);
/* Both line terms of VolSDFLoss.forward (loss_wfr.py:52-65) in one launch: pred_px / pred_calib [R,4] = the projected 3-D lines in
 * pixel and in calibrated coordinates, gt5 [R,5] = (x1, y1, x2, y2, weight), K [3,3].  out3 = (l2d pixel term, calibrated line
 * loss, #segments the pixel term accepts); the ground-truth end points are calibrated with K^-1 inside; d_pred_calib [R,4] =
 * grad_scale * d out3[1] / d pred_calib (grad_scale, ABI v12: the line term's weight in the total loss, or 1).  Same arithmetic as
 * neat_line_loss + neat_inv_small + neat_project2d on the same inputs. */
int neat_line_losses(const float* pred_px, const float* pred_calib, const float* gt5, const float* K, int R, float threshold, float* out3,
                     float* d_pred_calib, float grad_scale, void* stream);

/* ---- a16: Adam step over one flat fp32 parameter buffer = torch.optim.Adam(lr) as the reference trainer builds it
 * (training/volsdf_train.py:177; no weight decay, no amsgrad).  The parameters and both moments are flat [n]; the
 * gradients stay where autograd left them: grads[s] (device pointer, or NULL = no gradient this step: that segment is
 * skipped like torch does) covers elements [seg_offsets[s], seg_offsets[s+1]); seg_steps[s] = 1-based count of the steps
 * that segment has taken including this one (torch counts steps per tensor) -- host arrays, nseg <= 96. */
int neat_adam_step(float* params, const float* const* grads, const long long* seg_offsets, const int* seg_steps, int nseg,
                   float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps, void* stream);
/* ABI v13: neat_adam_step with its step-dependent numbers in DEVICE memory -- coef [2 nseg] = (lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t))
 * per segment -- so that the launch can be captured in the step's HIP graph and replayed; grads[s] == NULL: segment untouched. */
int neat_adam_step_coef(float* params, const float* const* grads, const long long* seg_offsets, int nseg, float* exp_avg, float* exp_avg_sq,
                        const float* coef, float beta1, float beta2, float eps, void* stream);

/* ---- 8f-2 (next row): device-side rectangular assignment = scipy.optimize.linear_sum_assignment as called at
 * model/networks/neat_wfr_rend_a.py:473 and model/networks/loss_wfr.py:108 (same algorithm, float64 duals, same tie
 * rule => same assignment), without the host round trip.  cost [nr,nc] float32 row-major; row_mask [nr] bytes or NULL
 * (rows with 0 do not take part: replaces the reference's `[good]` compaction); outputs row_ind/col_ind
 * [min(nr,nc)] int64 sorted by row and padded with -1, *n_match = number of pairs (-1 if a cost is not finite). */
size_t neat_lsap_ws_bytes(int nr, int nc);
int neat_lsap(const float* cost, int nr, int nc, const unsigned char* row_mask, const unsigned char* col_mask, long long* row_ind,
              long long* col_ind, int* n_match, void* ws, void* stream);
/* (col_mask [nc] bytes or NULL: columns with 0 do not take part -- padded candidate sets such as the DBSCAN centres below) */

/* DBSCAN(eps, min_samples = 2) + cluster means of n <= 8192 3-D points = VolSDFNetwork.cluster_dbscan
 * (model/networks/neat_wfr_rend_a.py:328-339; sklearn on the host in the reference).  centres: (n/2) x 3 floats followed
 * by n/2 ints of scratch; clusters in sklearn's order (by first member), padded; valid [n/2] bytes; *count = clusters. */
size_t neat_dbscan_ws_bytes(int n);
int neat_dbscan_means(const float* points, int n, double eps, float* centres, unsigned char* valid, int* count, void* ws, void* stream);

/* ---- a9 alone: volume_rendering :540-554 given sdf [R,S] -> weights [R,S] (used by tests) -------- */
int neat_volume_weights(const float* z, const float* sdf, int R, int S, const float* beta, float* weights, void* stream);

/* ---- measurement support (bench.py): when enabled, every launch of the two GEMM-class kernels is bracketed by
 * HIP events on the caller's stream.  class 0 = layer_kernel (all MLP chains), class 1 = wgrad_kernel,
 * class 2 = the fused SDF primal chain (16-bit builds: PE + lin0..lin8 of the SDF network in one launch), class 3 = the fused
 * SDF adjoint chain (the normals), class 4 = the heads' fused chains (forward and backward, one launch per head each).
 * neat_prof_collect synchronises on the recorded events and returns the summed kernel time, the summed
 * ALGORITHMIC flops (2*N*K*P with the true layer dims), the summed ALGORITHMIC HBM bytes (each operand/result row once,
 * weights once) and the launch count since neat_prof_enable(1). */
/* A/B switches for benchmarking and for the cross-check tests (tests/test_gpu_parity.py::test_bf16_*_kernels_agree); every
 * setting computes the same result up to summation order.  Returns -1 for an unknown key / value.
 *   0 bf16 layer-kernel point tile (2 = 64 points, 4 = 128)   1 weight gradient: 1 = tr16 streaming kernel, 0 = previous
 *   2 hidden layers: 1 = weight-stationary streaming kernel   3 persistent workgroups of that kernel (default 256)
 *   4 fused primal chain: 1 = weight-stationary               5 its batch in 32-point tiles (0 = auto, 2..4)
 *   6 partial reduction: 2 = one launch, 16-byte loads (default), 0 = group sums + finish, 1 = one 16-wave pass
 *   7 interleave weight gradients with the reverse chain (default 0)
 *   8 same-shaped weight gradients per launch (-1 = by problem size (default), 0 = one layer per launch, 2, 3, 6)
 *   9 point tiles of the persistent streaming kernels: 1 = interleaved over the workgroups (default), 2 = interleaved with an
 *     XCD-contiguous slot order (measured neutral), 0 = one contiguous range each
 *  10 fused primal chain: batches interleaved over the workgroups (default 0)
 *  11 non-temporal accesses, bit mask (default 15): 1 / 2 = aux0 / aux1 fetch of the layer kernels, 4 = weight-gradient operands,
 *     8 = `in` fetch of the layer kernels, 16 = out1 (m_l) store, 32 = out0 store (both stores measured neutral)
 *  12 layer kernels: 1 = 16-byte output stores through v_permlane32_swap (default; measured neutral), 0 = 8-byte stores
 *  13 16-bit builds: the adjoint chain (normals) as one fused launch (default 1), 0 = seed + eight streaming launches
 *  14 16-bit builds: the two heads as fused chains (kernels_heads.hpp): 2 = forward and backward (default), 1 = forward only,
 *     0 = one layer_kernel_ws launch per layer
 *  15 with the fused head backward: 1 = a head's weight gradients right after its backward chain (default: its cotangents are the
 *     last 270 MB written, -0.025 ms per step), 0 = after both chains, the two heads' hidden layers batched together
 *  16 16-bit builds: weight gradients of the SDF layers 1..7 contracted inside the tangent / reverse launches that hold both
 *     operands in LDS (kernels_dw.hpp): 1 = from 49 152 points on (default), 2 = always, 0 = never (separate launches)
 *  17 probe switch of those launches (4 = no partial stores: wrong results)      18 sub-ranges of their gather launch (default 16)
 *  19 16-bit builds: the heads' input layers ([256 feature | <= 64 small] columns) as ONE five-column-block weight-gradient launch
 *     (default 1; 0 = two launches that each read the whole cotangent array)
 *  20 a head's output-layer weight gradient as a fourth problem of its hidden layers' launch (default 1)
 *  21 16-bit builds: lin0's weight gradient (K = 39 PE columns) on the one-column-block variant of the streaming kernel: 0 = off,
 *     n = 1..4: on, with n times the point splits (default 1; 2 and 4 measured no faster)
 *  22 with key 16: the feature rows of lin8's weight gradient contracted inside lin8's reverse launch as well (default 1)
 *  23 workgroups (= partials per set) of the launches of key 16, 16..256 (default 256; C4's shape: 192 the same, 128 +7 %)
 *  24 with keys 16 and 22: the chain variables of those launches (tangents of h_2..h_7, cotangents of a_6..a_1) alternate between two
 *     buffers each instead of one array per layer (default 1; results bit-identical, fewer HBM write-backs)
 *  25 with key 16: consecutive layers of the tangent / reverse chains that share an epilogue variant run as ONE launch in which every
 *     workgroup loops over the layers for its own tiles (tangent 1-2 | 3 | 4-7, reverse 8 | 7-5 | 4 | 3-1: 15 launches -> 7; default 1;
 *     results bit-identical)
 *  28 the two 256 x 256 layers of the global-junction MLP (neat_ffn_forward / neat_ffn_backward's data path) on the fp32 matrix pipe
 *     (v_mfma_f32_32x32x2_f32: exact fp32 products, fixed summation order) instead of the vector-ALU kernel (default 1) */
int neat_set_tuning(int key, int value);
int neat_prof_enable(int on);
int neat_prof_collect(int cls, double* total_ms, double* total_flops, int* launches, double* total_bytes);

#ifdef __cplusplus
}
#endif
#endif
