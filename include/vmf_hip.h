/* vmf_hip.h -- C ABI of libvmf_hip.so: the MI355X (gfx950) implementation of the von Mises-Fisher
 * latent-sampling path of KHOUTAIBI/MedicalImageAnalysis.
 *
 * The reference is pure Python; the binding a maintainer adds is a ctypes stub in each of
 * sampler/von_mises_fisher.py, sampler/ive.py, sampler/spherical_uniform.py, model/spherical_vae.py,
 * model/torus_vae.py.  Every entry point below names the reference lines whose
 * arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the current HIP device unless marked "host";
 *   - the library never allocates, frees or keeps device memory: outputs and workspaces belong to the caller;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*) and never synchronises;
 *   - return value: 0 on success, a negative vmf_status otherwise; vmf_last_error_string() describes the
 *     last failure on the calling thread;
 *   - dtype selects the arithmetic AND storage type of every floating-point buffer of the call
 *     (VMF_F32 or VMF_F64), except where a parameter is declared double*;
 *   - tensors are dense row-major.
 *   - counter-based noise: Philox4x32-10 with key {seed_lo, seed_hi ^ offset_hi} and counter
 *     {index_lo, index_hi, slot, offset_lo}; `index` is the GLOBAL row / element index, so a batch sharded
 *     by rows over several GPUs draws the same numbers as the unsharded batch.  Every sampling entry point
 *     takes the host `offset` and an optional device pointer `offset_dev` (one uint64, may be NULL): the kernel
 *     uses offset + *offset_dev.  A launch captured into a HIP graph has its host arguments frozen; with the
 *     counter in device memory (advanced by the caller between launches) every replay draws fresh noise.
 */
#ifndef VMF_HIP_H
#define VMF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMF_HIP_VERSION 141

typedef void* vmf_stream_t; /* hipStream_t */

enum vmf_status {
  VMF_OK = 0,
  VMF_ERR_BAD_ARG = -1,     /* null pointer, negative size, inconsistent shapes */
  VMF_ERR_DTYPE = -2,       /* dtype is neither VMF_F32 nor VMF_F64 */
  VMF_ERR_LAUNCH = -3,      /* hipGetLastError() after the launch was not hipSuccess */
  VMF_ERR_UNSUPPORTED = -4  /* valid request this build has no kernel for (e.g. m > 2048) */
};

enum vmf_dtype { VMF_F32 = 0, VMF_F64 = 1 };

enum vmf_kl_variant {
  VMF_KL_SPHERICAL_PIECEWISE = 0, /* model/spherical_vae.py:141-184 (clamp, <1e-3 -> 0, >20 asymptote) */
  VMF_KL_TORUS_RAW = 1,           /* model/torus_vae.py:190-194 (unguarded sinh/cosh) */
  VMF_KL_CIRCULAR_EPS = 2,        /* CNN_experiment/circular_VAE.py:126-129 (sinh + 1e-6) */
  VMF_KL_S1_I0I1 = 3,             /* regular_cnn/s1_vae.py:83-88 (k I1/I0 - log I0, unscaled Bessel) */
  VMF_KL_STABLE_M3 = 4,           /* same quantity as 0-2 without overflow (no reference counterpart) */
  VMF_KL_STABLE_M2 = 5            /* same quantity as 3 without overflow (no reference counterpart) */
};

int vmf_version(void);
const char* vmf_last_error_string(void);

/* ---- VonMisesFisher3D ------------------------------------------------------------------------
 * Broadcast description shared by the two calls below.  The result of rsample has `ndim` leading dims
 * odims[] (then the trailing 3).  Output row o (row-major over odims) reads
 *     u / w   at element  sum_d idx_d * w_str[d]      (tensor of shape sample_shape + scale.shape)
 *     scale   at element  sum_d idx_d * k_str[d]
 *     loc     at row      sum_d idx_d * l_str[d]
 *     eps     at row      sum_d idx_d * e_str[d]      (tensor of shape sample_shape + loc.shape)
 * with stride 0 on broadcast dims.  odims and the four stride arrays are HOST arrays of length ndim
 * (1 <= ndim <= 8). */

/* flags of the rsample entry points (vmf3_rsample_*, vmf_rsample_*) */
#define VMF_FLAG_NORMALIZE_MU 1   /* general m: mu rows are raw (e.g. straight from a GEMM): sample around mu / ||mu||;
                                     the backward pass returns the gradient with respect to the RAW rows */
#define VMF_FLAG_SOFTPLUS_KAPPA 2 /* `scale` / `kappa` holds the raw output of the concentration head: the kernels use
                                     kappa = softplus(raw) + 1 (model/spherical_vae.py:70, model/torus_vae.py:103-104,
                                     CNN_experiment/circular_VAE.py:50; torch's softplus, beta 1, threshold 20) and
                                     the backward pass returns the gradient with respect to the RAW values */

/* Replaces VonMisesFisher3D.__init__'s normalisation and .rsample (sampler/von_mises_fisher.py:30, 50-86):
 *   l = loc/(||loc||+1e-5);  w = 1 + logsumexp(log u, log(1-u) - 2k)/k;  v = normalise(eps - <eps,l> l);
 *   z = w l + sqrt(clamp(1-w^2, 1e-10, 1)) v.
 * u / eps may be null: then both come from Philox(index, slot 0) = (x, y, z, w), 23 bits per uniform: u[i] from the
 * upper bits of x of index row0 + i; eps row j = Box-Muller(y, z) -> eps0, eps1 and Box-Muller(w, t) -> eps2 (cosine
 * branch) of index row0 + j, t being the lower 9 bits of x, y, z side by side.  row0 is the GLOBAL index of the call's
 * first u element / eps row: a rank that holds rows [a, b) of a batch passes row0 = a and draws exactly the rows the
 * unsharded call draws (both broadcast shapes with kernels of their own: (B,) and (B, 1) scale; with sample dimensions
 * in front it is a plain offset into the flattened u / eps tensors).  The backward entry points take the same row0.  w_out (nullable) receives w, indexed like u; kappa_out (nullable) receives the kappa the
 * kernel used, indexed like scale (of interest with VMF_FLAG_SOFTPLUS_KAPPA: the KL term needs it).
 * fp32 calls evaluate log / exp / sqrt / 1/x with the 1-ulp hardware instructions, fp64 calls with the accurate
 * library.  Two broadcast shapes have kernels of their own (vmf3_broadcast_kind tells which a call is): every operand
 * indexed by the output row ((B, 3) loc with (B,) scale), and the outer product u / scale x loc / eps that a (B, 1)
 * scale produces ((B, B, 3), model/spherical_vae.py:81-83). */
int vmf3_rsample_fwd(const void* loc, const void* scale, const void* u, const void* eps, void* z, void* w_out,
                     void* kappa_out, int ndim, const int64_t* odims, const int64_t* w_str, const int64_t* k_str,
                     const int64_t* l_str, const int64_t* e_str, int flags, int dtype, uint64_t row0, uint64_t seed, uint64_t offset,
                     const uint64_t* offset_dev, vmf_stream_t stream);

/* 1: every operand is indexed by the output row; 2: outer product (*n_i = elements of u / scale, *n_j = rows of
 * loc / eps; the result is (n_i, n_j, 3)); 0: any other broadcast.  Host only, touches no GPU.  n_i / n_j nullable. */
int vmf3_broadcast_kind(int ndim, const int64_t* odims, const int64_t* w_str, const int64_t* k_str,
                        const int64_t* l_str, const int64_t* e_str, int64_t* n_i, int64_t* n_j);

/* Backward of the above (what autograd derives from sampler/von_mises_fisher.py:30, 58-84):
 * dloc (n_loc rows of 3) and dscale (n_scale) receive J^T grad_z.  They are overwritten.  When several
 * output rows share a loc row or a scale element the partial results are combined with float atomics --
 * unless n_loc == n_scale == number of output rows: then row o of dloc / dscale receives output row o's
 * contribution and the caller reduces over the rows that share an operand (reproducible run to run).
 * grad_kappa (nullable, indexed like scale): gradient with respect to the kappa_out of the forward call, added to
 * dscale before the softplus chain rule; accepted for broadcast kind 1 only.  Kind 2 is better served by
 * vmf3_rsample_bwd_outer. */
int vmf3_rsample_bwd(const void* loc, const void* scale, const void* u, const void* eps, const void* grad_z,
                     const void* grad_kappa, void* dloc, void* dscale, int64_t n_loc, int64_t n_scale, int ndim,
                     const int64_t* odims, const int64_t* w_str, const int64_t* k_str, const int64_t* l_str,
                     const int64_t* e_str, int flags, int dtype, uint64_t row0, uint64_t seed, uint64_t offset,
                     const uint64_t* offset_dev, vmf_stream_t stream);

/* Backward of the outer-product shape (broadcast kind 2: grad_z is (n_i, n_j, 3), u / scale have n_i elements,
 * loc / eps n_j rows) in one pass over grad_z: dloc (n_j, 3) = sum over i, dscale (n_i) = sum over j, both reduced
 * inside the library in a fixed order (same bits every run, no (n_i, n_j, 3) temporary).  workspace: device memory
 * of at least the byte count vmf3_rsample_bwd_outer_workspace reports (about 3.4 % of grad_z for large shapes),
 * 16-byte aligned, contents irrelevant before and after the call.  grad_kappa as above. */
int vmf3_rsample_bwd_outer_workspace(int64_t n_i, int64_t n_j, int dtype, int64_t* bytes_out);
int vmf3_rsample_bwd_outer(const void* loc, const void* scale, const void* u, const void* eps, const void* grad_z,
                           const void* grad_kappa, void* dloc, void* dscale, int64_t n_i, int64_t n_j, void* workspace,
                           int64_t workspace_bytes, int flags, int dtype, uint64_t row0, uint64_t seed, uint64_t offset,
                           const uint64_t* offset_dev, vmf_stream_t stream);

/* _sample_w3 alone  sampler/von_mises_fisher.py:50-64: w[i] = 1 + logsumexp([log u_i, log(1-u_i) - 2k]) / k with
 * k = scale[i mod n_scale] (u, w have shape sample_shape + scale.shape, n elements); u null: Philox(index i, slot 0),
 * the same stream vmf3_rsample_fwd uses for its u.  bwd writes grad_w[i] * dw/dk per ELEMENT (n values); the caller
 * sums over the sample dimensions. */
int vmf3_sample_w_fwd(const void* scale, const void* u, void* w, int64_t n, int64_t n_scale, int dtype, uint64_t row0, uint64_t seed,
                      uint64_t offset, const uint64_t* offset_dev, vmf_stream_t stream);
int vmf3_sample_w_bwd(const void* scale, const void* u, const void* grad_w, void* dscale_rows, int64_t n,
                      int64_t n_scale, int dtype, uint64_t row0, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                      vmf_stream_t stream);

/* CNN_experiment/von_mises_fisher.py:88-108 with the reference's sign: logp = k<l,x> - lognorm,
 * lognorm = log k - log 4pi - log(sinh k + 1e-10).  logp / lognorm nullable. */
int vmf3_log_prob_ref(const void* loc, const void* scale, const void* x, void* logp, void* lognorm, int64_t n,
                      int dtype, vmf_stream_t stream);

/* Backward of vmf3_log_prob_ref (what autograd derives from CNN_experiment/von_mises_fisher.py:92-108): gl = dL/dlogp;
 * dloc (n,3), dscale (n), dx (n,3) nullable. */
int vmf3_log_prob_ref_bwd(const void* loc, const void* scale, const void* x, const void* gl, void* dloc, void* dscale,
                          void* dx, int64_t n, int dtype, vmf_stream_t stream);

/* VonMisesFisher3D.mean (sampler/von_mises_fisher.py:38-43): l * ive(3/2,k)/ive(1/2,k) */
int vmf3_mean(const void* loc, const void* scale, void* mean, int64_t n, int dtype, vmf_stream_t stream);

/* ---- ive --------------------------------------------------------------------------------------
 * Replaces IveFunction.forward / .backward (sampler/ive.py:12-63), which round-trip through scipy on the
 * host.  v is the (scalar) order.  Edge conventions follow scipy as the reference sees it: ive(v,0) = [v==0];
 * negative z: (-1)^v ive(v,|z|) for integer v, NaN otherwise; orders other than 0 and 1 return NaN for
 * |z| > 2^30 (AMOS's limit).  Regimes: power series / continued fraction (ratio) / uniform asymptotic. */
int vmf_ive_fwd(double v, const void* z, void* out, int64_t n, int dtype, vmf_stream_t stream);
/* gz = gout * (ive(v-1,z) - (v/z_safe + sign z) ive(v,z)),  z_safe = (|z|<1e-8 ? 1e-8 : z)   (:54-62) */
int vmf_ive_bwd(double v, const void* z, const void* gout, void* gz, int64_t n, int dtype, vmf_stream_t stream);
/* log ive(v,z) for z > 0 without underflow, and I_v(z)/I_{v-1}(z) (no reference counterpart; they feed the
 * general-m log-normaliser, entropy and KL). */
int vmf_log_ive_fwd(double v, const void* z, void* out, int64_t n, int dtype, vmf_stream_t stream);
int vmf_bessel_ratio_fwd(double v, const void* z, void* out, int64_t n, int dtype, vmf_stream_t stream);

/* ---- KL terms ---------------------------------------------------------------------------------
 * Per-element KL(vMF || uniform) in one of the reference's variants (enum vmf_kl_variant).
 *   out      nullable: per-element values (what kl_vmf_spherical_uniform returns)
 *   sum_out  nullable: one double = alpha * sum(out)  (alpha = 1/n gives the `.mean()` of
 *            model/spherical_vae.py:134); needs `workspace` of at least VMF_KL_WORKSPACE_DOUBLES doubles. */
#define VMF_KL_WORKSPACE_DOUBLES 1024
int vmf_kl_fwd(const void* kappa, void* out, double* sum_out, double* workspace, double alpha, int64_t n,
               int variant, int dtype, vmf_stream_t stream);
/* vmf_kl_fwd without the workspace argument.  No counterpart in the reference: it is for callers that cannot
 * allocate device memory -- a stream under HIP graph capture, a tight serving loop, a harness that forbids hipMalloc.
 * The block partials of the fused sum live in an array inside the library's code object (VMF_KL_WORKSPACE_DOUBLES
 * doubles = 8 KiB on every GPU the library runs on, until the process ends); the call allocates nothing.  Same
 * kernels, same launch geometry and therefore the same bits as vmf_kl_fwd.  n <= VMF_KL_STATIC_MAX_N, larger sizes
 * are refused (VMF_ERR_UNSUPPORTED).  Every call on one GPU shares that GPU's array: calls that may overlap there
 * (different streams) race on it -- order them, or give the overlapping ones vmf_kl_fwd with workspaces of their own.
 * Nothing in the array needs clearing between calls.  The first call on a device looks the array's address up
 * (hipGetSymbolAddress); make it before a capture begins.  The array used is the CURRENT device's: `stream` must
 * belong to the current device (a stream of another device is refused with VMF_ERR_BAD_ARG). */
#define VMF_KL_STATIC_MAX_N ((int64_t)1 << 30)
int vmf_kl_fwd_static(const void* kappa, void* out, double* sum_out, double alpha, int64_t n, int variant, int dtype,
                      vmf_stream_t stream);
/* dkappa[i] = gout_i * dKL/dkappa(kappa_i); gout is per element (gout_stride = 1) or one scalar
 * (gout_stride = 0), in the call's dtype; gscale multiplies it (1/n for the fused mean). */
int vmf_kl_bwd(const void* kappa, const void* gout, int64_t gout_stride, double gscale, void* dkappa, int64_t n,
               int variant, int dtype, vmf_stream_t stream);

/* ---- general-m von Mises-Fisher (no reference counterpart: the reference asserts m == 3) ----------
 * Wood (1994) rejection sampler for w with Ulrich's symmetric-Beta proposal, tangent direction from m-1
 * Gaussians, Householder reflection e1 -> mu:
 *     b = (m-1)/(2k + sqrt(4k^2+(m-1)^2)), a = ((m-1)+2k+sqrt(4k^2+(m-1)^2))/4, d = 4ab/(1+b) - (m-1)ln(m-1)
 *     proposal e ~ Beta((m-1)/2,(m-1)/2), u ~ U(0,1): w = (1-(1+b)e)/(1-(1-b)e), t = 2ab/(1-(1-b)e),
 *     accept iff (m-1) ln t - t + d >= ln u
 *     z' = [w, sqrt(max(1-w^2,1e-10)) g/||g||],  z = z' - 2<z',h>h/<h,h>,  h = e1 - mu.
 * mu is (n, m) with row stride mu_stride elements (0: one shared mean), kappa has stride kappa_stride (1 or 0).
 * Rows are numbered row0 .. row0+n-1 for the noise counters.
 * Noise: in-kernel Philox by default.  Pre-drawn mode (parity checks): prop_e, prop_u are (n, n_prop)
 * proposals and gauss is (n, m) normals (column 0 ignored); all three must then be given.
 * The accepted proposal index (first accepted, or the last one tried when none is) goes to accept_idx.
 * Outputs w_out (n), omw_out (n; 1 - w computed without cancellation, what the backward wants),
 * accept_idx (n, int32), logp_out (n; log p(z) fused), kappa_out (n; the kappa the kernel used, of interest with
 * VMF_FLAG_SOFTPLUS_KAPPA: the KL term needs it; kappa_stride must be 1 then) are nullable. */
/* flags: VMF_FLAG_NORMALIZE_MU, VMF_FLAG_SOFTPLUS_KAPPA (defined with vmf3_rsample_fwd above) */
int vmf_rsample_fwd(const void* mu, int64_t mu_stride, const void* kappa, int64_t kappa_stride, void* z, void* w_out,
                    void* omw_out, int32_t* accept_idx, void* logp_out, void* kappa_out, int64_t n, int m, const void* prop_e, const void* prop_u,
                    int n_prop, const void* gauss, int max_prop, uint64_t row0, uint64_t seed, uint64_t offset,
                    const uint64_t* offset_dev, int flags, int dtype, vmf_stream_t stream);
/* Backward through z given the saved w and 1-w (omw; null: 1-w is recomputed from w) -- path-wise through
 * b(kappa), the accepted proposal held fixed.
 * gauss null: the Gaussians are regenerated from the same counters.  dmu is (n, m); with mu_stride == 0 or
 * kappa_stride == 0 the caller reduces dmu / dkappa over rows.  grad_kappa (n, nullable): gradient with respect to the
 * forward call's kappa_out, added to dkappa before the softplus chain rule of VMF_FLAG_SOFTPLUS_KAPPA. */
int vmf_rsample_bwd(const void* mu, int64_t mu_stride, const void* kappa, int64_t kappa_stride, const void* w,
                    const void* omw, const void* grad_z, const void* gauss, const void* grad_kappa, void* dmu, void* dkappa, int64_t n, int m, uint64_t row0,
                    uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int flags, int dtype,
                    vmf_stream_t stream);
/* logp = k<mu,x> + log C_m(k),  log C_m = (m/2-1) log k - (m/2) log 2pi - (log ive(m/2-1,k) + k) */
int vmf_log_prob_fwd(const void* mu, int64_t mu_stride, const void* kappa, int64_t kappa_stride, const void* x,
                     void* logp, int64_t n, int m, int dtype, vmf_stream_t stream);
/* gl = dL/dlogp (n): dmu = gl k x, dx = gl k mu, dkappa = gl (<mu,x> - A_m(k)); outputs nullable */
int vmf_log_prob_bwd(const void* mu, int64_t mu_stride, const void* kappa, int64_t kappa_stride, const void* x,
                     const void* gl, void* dmu, void* dx, void* dkappa, int64_t n, int m, int dtype,
                     vmf_stream_t stream);
/* per-row entropy H = -log C_m(k) - k A_m(k), KL(vMF || U(S^{m-1})) = -H + log|S^{m-1}|, and their common
 * derivative magnitude dKL/dk = -dH/dk = k A_m'(k) (dkl_dk); every output nullable */
int vmf_entropy_kl_fwd(const void* kappa, void* entropy, void* kl, void* dkl_dk, int64_t n, int m, int dtype,
                       vmf_stream_t stream);

/* ---- uniform prior and S^1 sampler -----------------------------------------------------------------
 * SphericalUniform.sample (sampler/spherical_uniform.py:31-33): out = g/||g||, g (n, dim) normals; g null:
 * Philox(index row0+i, slot 0x20000 + chunk). */
int vmf_sphere_uniform_sample(const void* gauss, void* out, int64_t n, int dim, uint64_t row0, uint64_t seed,
                              uint64_t offset, const uint64_t* offset_dev, int dtype, vmf_stream_t stream);
/* S1_VAE.reparameterize (regular_cnn/s1_vae.py:48-55): z = normalize(mean*kappa + noise) with F.normalize's
 * eps 1e-12; mean/noise/z (n, dim), kappa (n); noise null: Philox as above with slot 0x30000 + chunk. */
int vmf_s1_projnormal_fwd(const void* mean, const void* kappa, const void* noise, void* z, int64_t n, int dim,
                          uint64_t row0, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int dtype,
                          vmf_stream_t stream);
int vmf_s1_projnormal_bwd(const void* mean, const void* kappa, const void* noise, const void* gz, void* dmean,
                          void* dkappa, int64_t n, int dim, uint64_t row0, uint64_t seed, uint64_t offset,
                          const uint64_t* offset_dev, int dtype, vmf_stream_t stream);

/* ---- the steps on either side of the sampler in the train step ------------------------------------------
 * ELBO  model/spherical_vae.py:119-138 (= model/torus_vae.py:160-180, CNN_experiment/circular_VAE.py:90-116,
 * and with sums instead of means regular_cnn/s1_vae.py:70-92):
 *   loss = a_rec * sum_e (x[e mod n_x] - x_hat[e])^2 + a_kl * sum_i KL_variant(kappa_i)
 * as one fused reduction.  x (n_x elements) repeats along the leading dimensions of x_hat (n_hat elements, a
 * multiple of n_x) the way `x - x_mu` broadcasts when the sample is (B, B, 3).  The caller passes
 * a_rec = gamma / n_hat and a_kl = beta / n_kappa for the reference's means.  loss_out: one element of `dtype`;
 * parts_out (optional): two doubles {a_rec * sum, a_kl * sum}; workspace: VMF_ELBO_WORKSPACE_DOUBLES doubles.
 * bwd: grad_loss is a DEVICE scalar of `dtype` (no host sync); d_x_hat / d_kappa may be NULL to skip. */
#define VMF_ELBO_WORKSPACE_DOUBLES 2048
int vmf_elbo_fwd(const void* x, const void* x_hat, const void* kappa, void* loss_out, double* parts_out,
                 double* workspace, double a_rec, double a_kl, int64_t n_x, int64_t n_hat, int64_t n_kappa,
                 int variant, int dtype, vmf_stream_t stream);
int vmf_elbo_bwd(const void* x, const void* x_hat, const void* kappa, const void* grad_loss, void* d_x_hat,
                 void* d_kappa, double a_rec, double a_kl, int64_t n_x, int64_t n_hat, int64_t n_kappa, int variant,
                 int dtype, vmf_stream_t stream);

/* Torus embedding  model/torus_vae.py:60-81: rows of z_theta / z_phi (n, zdim >= 2) hold (cos, sin) in their
 * first two components; out (n, 3) = ((R + r cos t) cos p, (r + r cos t) sin p, r sin t)  (y as in :78).
 * bwd writes full (n, zdim) gradients (components >= 2 are zero). */
int vmf_build_torus_fwd(const void* z_theta, const void* z_phi, void* out, int64_t n, int zdim, double R, double r,
                        int dtype, vmf_stream_t stream);
int vmf_build_torus_bwd(const void* z_theta, const void* z_phi, const void* grad_out, void* g_theta, void* g_phi,
                        int64_t n, int zdim, double R, double r, int dtype, vmf_stream_t stream);

/* Supervised latent losses  dataloader/utils.py:101-180.  z (n, zdim) row-major, labels (n_labels, label_cols);
 * row i of z pairs with label row i mod n_labels (z may carry leading broadcast dimensions).  mean_out: one
 * element of `dtype` = mean over the n rows; per_row_out (optional): n elements; workspace:
 * VMF_KL_WORKSPACE_DOUBLES doubles.  R, r are used by VMF_LL_T2 only.  bwd: grad_mean is a device scalar. */
enum vmf_latent_loss_variant {
  VMF_LATENT_S1 = 0, /* (1 - cos(atan2(z1, z0) - label0))^2                                   :113-123 */
  VMF_LATENT_S2 = 1, /* (1 - cos(dtheta) + sin(theta_gt) sin(theta_hat) (1 - cos(dphi)))^2     :128-151 */
  VMF_LATENT_T2 = 2  /* (1 - cos dtheta) + (1 - cos dphi) + 0.1 ((rho - R)^2 + z^2 - r^2)^2    :153-177 */
};
int vmf_latent_loss_fwd(const void* z, const void* labels, void* per_row_out, void* mean_out, double* workspace,
                        int64_t n, int zdim, int64_t n_labels, int label_cols, int variant, double R, double r,
                        int dtype, vmf_stream_t stream);
int vmf_latent_loss_bwd(const void* z, const void* labels, const void* grad_mean, void* dz, int64_t n, int zdim,
                        int64_t n_labels, int label_cols, int variant, double R, double r, int dtype,
                        vmf_stream_t stream);

/* ---- multi-GPU ---------------------------------------------------------------------------------------
 * The path shards by row and exchanges nothing but loss scalars: the `.mean()` / `.sum()` of
 * model/spherical_vae.py:134,136 and regular_cnn/s1_vae.py:75,90 taken over all ranks' rows.  buf (device, `count`
 * doubles, e.g. [sum KL, sum reconstruction, sum log p, rows]) is summed over the ranks of `comm` (an ncclComm_t of
 * the RCCL library loaded in the process) in place, asynchronously on `stream`; divide by the summed row count
 * afterwards.  The reference is single-device (model/train.py:36-37) and has no counterpart. */
#define VMF_ALLREDUCE_MAX_SCALARS 64
int vmf_allreduce_scalars(void* comm, double* buf, int count, vmf_stream_t stream);

/* ---- device-resident noise counter ----------------------------------------------------------------------
 * Every sampling entry point takes `offset_dev`, a device-resident 64-bit counter added to `offset` inside the kernel
 * (the host arguments of a launch captured into a HIP graph are frozen; with the counter on the device every replay
 * draws fresh noise).  One sampling call = one ticket: snapshot[0] = counter[0]; counter[0] += 1 -- in ONE launch on
 * `stream` (the two tensor operations it replaces were two).  The snapshot is what the call's forward and backward
 * pass as `offset_dev`.  The reference draws from torch's global generator (sampler/von_mises_fisher.py:56,75) and
 * has no counterpart. */
int vmf_philox_ticket(int64_t* counter, int64_t* snapshot, vmf_stream_t stream);

/* Every compute entry point opens a marker range of its own name when the environment has VMF_ROCTX=1
 * (`rocprofv3 --marker-trace`); no other observable state. */

#ifdef __cplusplus
}
#endif
#endif /* VMF_HIP_H */
