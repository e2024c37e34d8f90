/* shw.h -- C ABI of the MI355X (gfx950) spherical sliced-Wasserstein / Chamfer hot path.
 *
 * The reference has no FFI layer: its "operator API" for this path is a set of dependency-injected
 * Python callables.  The entry points below are what those callables bind to;
 * each one names the reference code it replaces (paths relative to
 * /root/reference/Point_Cloud_Resistration/losses/).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 *   - the caller owns and allocates every buffer, including workspaces (sizes from the *_bytes
 *     helpers); nothing is allocated, freed or synchronised inside;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *   - return value is a hipError_t cast to int: 0 = success, 1 (hipErrorInvalidValue) = bad
 *     arguments / unsupported size, anything else = the launch error.
 *   - clouds are fp32 row-major (pairs, points, 3); directions are fp32 (slices, 3, 2) orthonormal
 *     2-frames, either one set per pair (u_pair_stride = slices*6) or shared (u_pair_stride = 0).
 */
#ifndef SHW_H
#define SHW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: these entry points are its only dynamic symbols */
#define SHW_API __attribute__((visibility("default")))

#define SHW_ABI_VERSION 3 /* 2: shw_ssw_backward_points takes per-pair upstream weights; 3: shw_circle_ot takes `method`,
                             shw_sinkhorn_forward_train takes plan / cost_matrix */
#define SHW_MAX_POINTS 8192 /* per cloud, per pair */

/* ABI version of the loaded library (== SHW_ABI_VERSION of the header it was built from). */
SHW_API int shw_abi_version(void);

/* Largest point count per cloud the sort kernels accept (SHW_MAX_POINTS). */
SHW_API int shw_max_points(void);

/* ---------------------------------------------------------------------------------------------
 * Direction frames.
 * Replaces: `U, _ = torch.linalg.qr(Z)` (max_spherical_sliced_w.py:308, _fast.py:318) for Z of shape (count, 3, 2):
 * reduced QR by Householder reflectors with LAPACK's sign convention, one thread per matrix.  The Gaussian
 * draw itself (`torch.randn`, :307) stays with the caller so the generator is consumed as in the reference.
 *   z (count, 3, 2) fp32 in, u (count, 3, 2) fp32 out (orthonormal columns).
 */
SHW_API int shw_stiefel_frames(const float* z, long count, float* u, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Spherical sliced-Wasserstein, forward.
 * Replaces: sliced_cost (max_spherical_sliced_w.py:251-286; batched _fast.py:258-295) =
 *   projection (:270-271) + normalise (:274-275) + circle coordinate (:278-279) + per-slice sort
 *   (:163-164 / :224-225) + circular OT solve (binary_search_circle :117-207 for p != 1,
 *   emd1D_circle :210-247 for p == 1) -- everything up to, not including, the mean over slices.
 *
 *   xs (pairs, n, 3), xt (pairs, m, 3), dirs (see header comment), p >= 1.
 *   slice_cost  (pairs*slices) fp32 out : circular OT cost W_p^p of every (pair, slice).
 *   slice_shift (pairs*slices) int32 out, may be NULL : optimal cyclic shift k* of the sorted
 *                target against the sorted source (p != 1), or the median level (p == 1).
 * This entry point: uniform weights; n == m for p != 1; any n, m for p == 1; 1 <= n, m <= SHW_MAX_POINTS.
 * (n != m or weights with p != 1: shw_ssw_forward_general.)
 */
SHW_API int shw_ssw_forward(const float* xs, const float* xt, const float* dirs,
                    int pairs, int n, int m, int slices, long u_pair_stride, float p,
                    float* slice_cost, int32_t* slice_shift, void* stream);

/* Reduction of per-slice costs to the reference's scalars.
 * Replaces: torch.mean(w1) (:286) per pair and the `w1 += mean(...)` loop over the batch
 * (_fast.py:291-293).
 *   pair_loss (pairs) out : scale * sum_l slice_cost[b, l]   (scale = 1/slices on one GPU, or
 *                           1/global_slices when slices are sharded across ranks)
 *   total     (2)     out : total[0] = sum_b pair_loss[b], total[1] = total[0] / pairs
 * Deterministic (fixed-order shuffle + serial tree, no atomics).
 */
SHW_API int shw_ssw_reduce(const float* slice_cost, int pairs, int slices, float scale,
                   float* pair_loss, float* total, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Spherical sliced-Wasserstein, forward + gradient in one pass.
 * Replaces: autograd through Cost -> gather -> sort -> atan2 -> normalize -> matmul
 * (max_spherical_sliced_w.py:207, :100-112, :163-164, :270-279); used by the notebooks' gradient
 * flow (Flow_cube.ipynb:1381-1383) and by any trainer that back-propagates through the loss.
 *
 * Computes slice_cost / slice_shift as shw_ssw_forward and, additionally,
 *   coef_s (pairs*slices*n) fp32 scratch : d cost(b,l) / d coord_s[b,l,i]  in ORIGINAL point order
 *   coef_t (pairs*slices*m) fp32 scratch : d cost(b,l) / d coord_t[b,l,j]
 * which shw_ssw_backward_points turns into d(sum_b pair_loss[b]) / d xs, / d xt:
 *   grad_xs[b,i,:] = scale * sum_l coef_s[b,l,i] * (-b_ U[:,0] + a_ U[:,1]) / (2 pi (a_^2 + b_^2)),
 *   (a_, b_) = U_l^T xs[b,i]   (SURVEY.md 8a row A9), likewise for xt.
 * Deterministic: every gradient element is summed over slices in a fixed order by one thread.
 */
SHW_API size_t shw_ssw_coef_bytes(int pairs, int n, int m, int slices);

SHW_API int shw_ssw_forward_grad(const float* xs, const float* xt, const float* dirs,
                         int pairs, int n, int m, int slices, long u_pair_stride, float p,
                         float* slice_cost, int32_t* slice_shift,
                         float* coef_s, float* coef_t, void* stream);

/* pair_w (pairs) and total_w (1) fp32, each may be NULL: upstream gradients d loss / d pair_loss[b] and
 * d loss / d total[0] of shw_ssw_reduce's outputs; row b of both gradients is multiplied by
 * pair_w[b] + total_w[0] inside the kernel (a NULL term counts as 0; both NULL = 1).  Any number of pairs
 * (batches beyond 65535 pairs go out as several launches). */
SHW_API int shw_ssw_backward_points(const float* xs, const float* xt, const float* dirs,
                            const float* coef_s, const float* coef_t,
                            int pairs, int n, int m, int slices, long u_pair_stride, float scale,
                            const float* pair_w, const float* total_w, float* grad_xs, float* grad_xt, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Spherical sliced-Wasserstein, general circular OT: n != m and / or non-uniform weights.
 * Replaces: binary_search_circle with u_weights / v_weights and unequal sample counts
 * (max_spherical_sliced_w.py:117-207, dCost :25-65, Cost :68-113), reached from sliced_cost (:284) when the
 * trainers use different source / target densities (train_W_COS.py:292-293,334-336) or the caller passes
 * weights (:289).  Follows the reference's bisection over the cut theta and its tangent exit.  p == 1 takes the
 * weighted form of the level-median formula (emd1D_circle, :210-247); slice_theta then holds the median level.
 *   wu (n) or (pairs, n), wv (m) or (pairs, m): non-negative weights summing to 1, NULL = uniform;
 *   w*_pair_stride = 0 when one weight vector is shared by all pairs (the reference's usage), else n / m.
 *   slice_theta (pairs*slices) fp32 out, may be NULL : the cut the solve ended on.
 *   coef_s / coef_t as in shw_ssw_forward_grad, both NULL for a loss-only evaluation.
 * 1 <= n, m <= 4096 on this path.
 */
SHW_API int shw_ssw_forward_general(const float* xs, const float* xt, const float* dirs,
                            const float* wu, const float* wv, long wu_pair_stride, long wv_pair_stride,
                            int pairs, int n, int m, int slices, long u_pair_stride, float p,
                            float* slice_cost, float* slice_theta, float* coef_s, float* coef_t, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Circle level: optimal transport between rows of circle COORDINATES (numbers in [0, 1]), no projection.
 * Replaces: binary_search_circle(u_values, v_values, u_weights, v_weights, p) (max_spherical_sliced_w.py:117-207)
 * and emd1D_circle(u_values, v_values, u_weights, v_weights) (:210-247), called on (rows, n) / (rows, m) coordinate
 * tensors -- the same kernels as the sliced entry points, their loaders reading one float per atom instead of
 * projecting a point on a frame.
 *   method : SHW_CIRCLE_BISECTION    = binary_search_circle for every p >= 1.  p == 1 is that function's DEFAULT (:117): the
 *                                      bisection over the cut ending in Cost's p == 1 branch (:107-108), i.e. the true circular
 *                                      W_1 -- NOT the value of emd1D_circle, whose formula leaves out the wrap segment;
 *            SHW_CIRCLE_LEVEL_MEDIAN = emd1D_circle (p must be 1);
 *            SHW_CIRCLE_AS_SLICED    = sliced_cost's own dispatch (:281-284): level-median for p == 1, bisection otherwise.
 *   u (rows, n), v (rows, m); wu / wv weights (n) / (m) shared (stride 0) or per row (stride n / m), NULL = uniform;
 *   cost (rows) out : W_p^p of every row;
 *   aux  (rows) out, may be NULL : int32 optimal shift k* (equal sizes, p != 1) or median level (p == 1), or the
 *                fp32 cut the bisection ended on (n != m or weights);
 *   grad_u (rows*n), grad_v (rows*m) out, both NULL for a value-only call : d cost[row] / d u[row, i], / d v[row, j].
 * Same size limits as the sliced entry points (8192; 4096 with weights or n != m and p != 1).
 */
#define SHW_CIRCLE_AS_SLICED 0
#define SHW_CIRCLE_BISECTION 1
#define SHW_CIRCLE_LEVEL_MEDIAN 2
SHW_API int shw_circle_ot(const float* u, const float* v, const float* wu, const float* wv, long wu_row_stride,
                          long wv_row_stride, int rows, int n, int m, float p, int method, float* cost, float* aux,
                          float* grad_u, float* grad_v, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Euclidean sliced-Wasserstein (the notebooks' SWD baseline).
 * Replaces: sliced_wasserstein_distance (Wasserstein_flow_problem/Flow_cube.ipynb:280-292): projection on unit
 * directions, per-slice sort of both projected sequences, sum of |sorted difference|^p.
 *   xs, xt (pairs, n, 3) -- equal counts, as the notebook code requires; thetas (slices, 3) shared
 *   (theta_pair_stride = 0) or (pairs, slices, 3) (stride = slices*3); p >= 1; n <= 4096; pairs <= 65535
 *   (pairs ride on gridDim.y in the two backward kernels).
 *   slice_sum (pairs*slices) out : S_l = sum_i |u_(i) - v_(i)|^p     (the notebook's outer (mean_l S_l)^(1/p)
 *                                  is host arithmetic on `slices` numbers)
 *   coef_s / coef_t (pairs*slices*n) scratch, both NULL for a value-only call : d S_l / d projection in
 *   original point order; shw_esw_backward_points turns them into
 *   grad_x[b,i,:] = sum_l slice_w[b,l] * coef[b,l,i] * theta[b,l,:]  (slice_w = upstream gradient of S).
 */
SHW_API int shw_esw_forward(const float* xs, const float* xt, const float* thetas, int pairs, int n, int slices,
                    long theta_pair_stride, float p, float* slice_sum, float* coef_s, float* coef_t, void* stream);

SHW_API int shw_esw_backward_points(const float* thetas, const float* coef_s, const float* coef_t, const float* slice_w,
                            int pairs, int n, int slices, long theta_pair_stride,
                            float* grad_xs, float* grad_xt, void* stream);

/* Gradient w.r.t. the directions (max_sliced_wasserstein_distance, Flow_cube.ipynb:294-323, ascends on them):
 *   grad_thetas[b,l,:] = slice_w[b,l] * sum_i (coef_s[b,l,i] * xs[b,i,:] + coef_t[b,l,i] * xt[b,i,:]),
 * always (pairs, slices, 3); directions shared by several pairs sum their rows on the host side. */
SHW_API int shw_esw_backward_dirs(const float* xs, const float* xt, const float* coef_s, const float* coef_t,
                          const float* slice_w, int pairs, int n, int slices, float* grad_thetas, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Log-domain Sinkhorn distance (comparison metric); value-only entry point.
 * Replaces: log_Sinkhorn_Distance_Loss.forward and log_N_Sinkhorn_Distance_Loss.forward
 * (/root/reference/Comparison_Wasserstein_with_Chamfer_distance/losses/sinkhorn.py:14-63, :104-157), called at
 * main_rotation.py:207-211.
 *   x (pairs, n, 3), y (pairs, m, 3); eps > 0; max_iter >= 0; norm_p = p of the coordinate-wise cost
 *   sum_d |x_d - y_d|^p ('L2' -> 2); cost_pow = N of the log_N variant (1 for the plain class);
 *   thresh = the convergence threshold on mean_b sum_i |u - u_old| (the reference hard-codes 1e-9).
 *   pairs <= 65535 (gridDim.y).
 *   workspace : shw_sinkhorn_workspace_bytes(pairs, n, m) bytes of device memory (duals, statistics);
 *               after the call its first pairs*n floats hold u and the next pairs*m floats hold v.
 *   cost (pairs) out : sum_ij exp(M_ij) C_ij  (before the batch reduction and the 1/N power, host side);
 *   plan, cost_matrix : optional dense (pairs, n, m) outputs P and C that the reference returns; NULL to skip.
 * The cost matrix is never stored unless asked for; the iteration count is fixed at enqueue time and the
 * convergence test is a device-side flag that turns the remaining launches into no-ops (no host sync).
 */
SHW_API size_t shw_sinkhorn_workspace_bytes(int pairs, int n, int m);

SHW_API int shw_sinkhorn_forward(const float* x, const float* y, int pairs, int n, int m, float eps, int max_iter,
                         int norm_p, int cost_pow, float thresh, void* workspace, float* cost, float* plan,
                         float* cost_matrix, void* stream);

/* Log-domain Sinkhorn with gradients w.r.t. both clouds.
 * Replaces: autograd through log_Sinkhorn_Distance_Loss.forward (sinkhorn.py:35-49: the iterations are differentiable).
 * shw_sinkhorn_forward_train runs the same iterations and keeps the trajectory of the duals (u_t, v_t), t = 0..T, in
 * `workspace` (shw_sinkhorn_train_workspace_bytes(pairs, n, m, max_iter) bytes: 2 (max_iter + 2) pairs (n + m) floats);
 * shw_sinkhorn_backward walks it from t = T down to 1, two kernels per iteration, recomputing the transport weights
 * from the points (nothing dense is stored), every gradient row owned by one thread (deterministic):
 *   grad_cost (pairs) in : upstream gradient of cost[b];  grad_x (pairs, n, 3), grad_y (pairs, m, 3) out (overwritten).
 * The workspace must be passed unchanged from the forward to the backward call (same sizes, eps, max_iter, norms).
 * plan, cost_matrix (ABI 3): optional dense (pairs, n, m) outputs P and C of the SAME solve (NULL to skip), plain values: the
 * gradient this library computes is that of `cost`.
 */
SHW_API size_t shw_sinkhorn_train_workspace_bytes(int pairs, int n, int m, int max_iter);

SHW_API int shw_sinkhorn_forward_train(const float* x, const float* y, int pairs, int n, int m, float eps, int max_iter,
                                       int norm_p, int cost_pow, float thresh, void* workspace, float* cost, float* plan,
                                       float* cost_matrix, void* stream);

SHW_API int shw_sinkhorn_backward(const float* x, const float* y, int pairs, int n, int m, float eps, int max_iter,
                                  int norm_p, int cost_pow, void* workspace, const float* grad_cost, float* grad_x,
                                  float* grad_y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Chamfer distance (comparison baseline).
 * Replaces: pytorch3d.loss.chamfer_distance with default arguments, as called at
 * train_CD.py:123,161,327-328, main_rotation.py:203, test_ERROR.py:216 (third-party arithmetic,
 * un-vendored and un-pinned: see DESIGN.md "parity unpinned").
 *   x (pairs, n, 3), y (pairs, m, 3)
 *   min_xy (pairs*n) fp32 out : min_j |x_i - y_j|^2 ;  nn_xy (pairs*n) int32 out : its argmin j
 *   min_yx (pairs*m) fp32 out : min_i |x_i - y_j|^2 ;  nn_yx (pairs*m) int32 out : its argmin i
 *   pair_loss (pairs) out : mean_i min_xy[b,i] + mean_j min_yx[b,j]   (deterministic reduction)
 * All five outputs are required (the index arrays feed shw_chamfer_backward).  pairs <= 65535 (gridDim.y).
 */
SHW_API int shw_chamfer_forward(const float* x, const float* y, int pairs, int n, int m,
                        float* min_xy, int32_t* nn_xy, float* min_yx, int32_t* nn_yx,
                        float* pair_loss, void* stream);

/* grad of sum_b w[b]*pair_loss[b] (w = per-pair upstream gradient, device pointer, (pairs)).
 * grad_x, grad_y are overwritten.  Owner-computed (round 3): every row is summed by one thread in a fixed order --
 * bit-identical from run to run (round 2 scattered with float atomics and needed zero-filled outputs). */
SHW_API int shw_chamfer_backward(const float* x, const float* y, const int32_t* nn_xy, const int32_t* nn_yx,
                         const float* w, int pairs, int n, int m,
                         float* grad_x, float* grad_y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SHW_H */
