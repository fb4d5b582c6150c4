/* fneus.h -- C ABI of libfneus_hip.so: the MI355X (gfx950) implementation of the Factored-NeuS stage-1
 * volume-rendering hot path.
 *
 * The reference (yiqun-wang/Factored-NeuS) has no FFI / operator registry: its hot path is eager PyTorch behind
 * the Python class API  models.renderer.NeuSRenderer  (renderer.py:80-500) and  models.fields.*  (fields.py).
 * Each entry point below replaces one group of reference ATen op sequences (cited per function); the Python host
 * side (factored-neus_amd/models/, factored-neus_amd/fneus/) binds them with ctypes and keeps the reference's class
 * API.  INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless noted; the caller owns all memory (no allocation inside);
 *   - `stream` is a hipStream_t (passed as void*); kernels are enqueued, never synchronised;
 *   - return 0 = ok, <0 = error (-1 launch failure, -2 bad argument);  fneus_last_error() gives a message;
 *   - `prec` selects the MFMA numerics: 1 = bf16 operands (fast), 3 = split-bf16 x3 (hi*hi+hi*lo+lo*hi, ~fp32
 *     parity mode); accumulation is always fp32;
 *   - network "blobs" are produced by fneus_pack() from flat fp32 parameter buffers (natural layout:
 *     per layer W[out][in] row-major then b[out]) according to a job table built by the host.
 *   - points can be given explicitly (`pts` [n][3]) or as rays: p = rays_o[n/m] + rays_d[n/m] * t[n]  (pts == NULL).
 */
#ifndef FNEUS_H
#define FNEUS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fneus_stream_t; /* hipStream_t */

/* Activation stash of the SDF network, written by fneus_sdf_fwd_grad and consumed by fneus_sdf_bwd / fneus_dw_gemm_pp.
 * h, a, pe are FRAGMENT PLANES (csrc/fneus_pp.h): per 32-sample tile the MFMA B fragments of a layer (1 KiB each = 16
 * features x 32 samples), bf16; tiles = ceil(N / 32).  The *_lo planes carry the bf16 remainder of every value: they are
 * optional (NULL = not written / not read; "gradient precision" 1) -- with them the weight gradients are fp32-accurate
 * (gradient precision 3). */
typedef struct FneusSdfStash {
    uint16_t* pe_hi;   uint16_t* pe_lo;   /* [tiles][4][512]      positional encoding (39 of 64 features; fragment 3 stays zero) */
    uint16_t* h_hi;    uint16_t* h_lo;    /* [8][tiles][16][512]  slot l = softplus output of layer l (= input of l+1)       */
    uint16_t* a_hi;    uint16_t* a_lo;    /* [8][tiles][16][512]  slot l = a_l = sigma'(z_l) * d sdf/d h_{l+1}  (train != 0)  */
    uint16_t* feat_hi; uint16_t* feat_lo; /* [tiles][16][512]     feature vector (colour-network input; train != 0)          */
    void* ps; /* sigma'(z_l) as 16-bit fixed point, lane-private: [tiles][8][16][64] x 16 bytes                              */
    float* qs; /* scratch of the reverse sweep (the skip input's part of d sdf/d PE), lane-private: [tiles][2][64][16] floats */
} FneusSdfStash;

/* Stash of the colour network / of one RefColor MLP (written by fneus_color_fwd with train != 0 and fneus_color_bwd):
 * fragment planes like FneusSdfStash, *_lo optional in the same way.  Gradient precision 2 (round 5): side_lo and zbar_lo NULL,
 * u_lo and zout_lo given -- of u_lo only slot 3 is then written (the other slots' addresses are never touched and need not
 * exist): the two operands of the output layer's weight gradient, the one product of the colour network whose bf16 rounding
 * exceeds the bounds of the exact mode. */
typedef struct FneusColStash {
    uint16_t* side_hi; uint16_t* side_lo; /* [tiles][4][512]      pts | PE4(view) | normal (33 of 64 features; fragment 3 zero) */
    uint16_t* u_hi;    uint16_t* u_lo;    /* [4][tiles][16][512]  slot l = ReLU output of layer l (= input of l+1)            */
    uint16_t* zbar_hi; uint16_t* zbar_lo; /* [4][tiles][16][512]  slot l = dL/dz_l                                             */
    uint16_t* zout_hi; uint16_t* zout_lo; /* [tiles][2][512]      dL/dz of the output layer (3 features)                       */
    void* mask;                           /* lane-private ReLU masks: [tiles][4][64] x 16 bytes                                */
    uint16_t* feat_hi; uint16_t* feat_lo; /* [tiles][16][512]     fneus_refcolor_* only: copy of the input features; NULL for
                                                                   the colour network (its features are FneusSdfStash.feat)    */
    uint16_t* dfeat_hi;                   /* [tiles][16][512]     fneus_color_bwd with d_feat == NULL (round 6): the feature
                                             cotangent as bf16 fragments -- slot 8 of FneusSdfBwdBufs.zbar_hi, where fneus_sdf_bwd
                                             (d_feat == NULL) takes the seed of its descending chain from; NULL otherwise     */
    int32_t dnormal_add;                  /* fneus_color_bwd, launches of >= 1024 sample tiles (-2 otherwise): non-zero = d_normal is
                                             ADDED to what the buffer holds (the compositing backward's gradient of the same normals:
                                             autograd's sum of the two, renderer.py:243 + fields.py:160, without a launch of its own) */
} FneusColStash;

/* work buffers of fneus_sdf_bwd: the operands of the weight-gradient GEMM (fragment planes like FneusSdfStash, *_lo
 * optional in the same way) + private scratch. */
typedef struct FneusSdfBwdBufs {
    uint16_t* qbar_hi; uint16_t* qbar_lo; /* [tiles][4][512]      adj_0 = J nbar (as pe)                                   */
    uint16_t* adj_hi;  uint16_t* adj_lo;  /* [8][tiles][16][512]  slot l = adj_{l+1}                                       */
    uint16_t* zbar_hi; uint16_t* zbar_lo; /* [9][tiles][16][512]  slot l = dL/dz_l (slot 8: the 256 feature rows of layer 8) */
    uint16_t* zsdf_hi; uint16_t* zsdf_lo; /* [tiles][2][512]      feature 0 = dL/dsdf (the sdf row of layer 8)             */
    uint16_t* c_hi;    uint16_t* c_lo;    /* coupling terms between the two chains, lane-private: [tiles][8][16][64] x 16 bytes */
} FneusSdfBwdBufs;

/* Activation planes of the background NeRF++ (fneus_nerf_bg_fwd / _bwd): FRAGMENT PLANES like every other stash since
 * round 2 (csrc/fneus_pp.h: [tiles][F fragments][64 slots][8 bf16], tiles = 2 ceil(n / 64), zero initialised), the operands
 * of the weight-gradient GEMM fneus_dw_gemm_pp.  The *_lo pointers are NULL unless the gradient precision is 3. */
typedef struct FneusNerfStash {
    void *pe_hi, *pe_lo;       /* F = 6   PE10 of the 4-D point, 84 features used                              */
    void *h_hi, *h_lo;         /* [8][tiles][16] slot l = relu output of pts_linears.l                         */
    void *feat_hi, *feat_lo;   /* F = 16  feature_linear output                                                */
    void *dpe_hi, *dpe_lo;     /* F = 2   PE4 of the view direction, 27 features used                          */
    void *hv_hi, *hv_lo;       /* F = 8   relu output of views_linears.0                                       */
    uint32_t* mask;            /* [tiles][9][64] x 4 words: ReLU sign bits, lane-private                       */
    void *zbar_hi, *zbar_lo;   /* [8][tiles][16] dL/dz of pts_linears.l              (written by the backward) */
    void *zfeat_hi, *zfeat_lo; /* F = 16  dL/d feature                                                         */
    void *zhv_hi, *zhv_lo;     /* F = 8   dL/dz of views_linears.0                                             */
    void *zout_hi, *zout_lo;   /* F = 4   fragments 0, 1: rows 0..2 = dL/d rgb; fragments 2, 3: row 0 = dL/d density */
} FneusNerfStash;

/* One product of fneus_dw_gemm_pp over FRAGMENT PLANES (csrc/fneus_pp.h): a plane holds, per 32-sample tile, the MFMA B
 * fragments of a layer's activations as the chain kernels produce them (1 KiB each = 16 features x 32 samples).
 * C[o][i] += scale * sum_tiles (A^T B + A2^T B2); bias[o] += sum_samples A[s][o].  *_lo planes (same layout) carry the
 * bf16 remainder of every value and are read in the exact-gradient mode (gprec 3) only. */
typedef struct FneusGemmPPJob {
    const void *a_hi, *a_lo, *b_hi, *b_lo;      /* plane bases (sample tile 0)                                     */
    const void *a2_hi, *a2_lo, *b2_hi, *b2_lo;  /* second term or NULL                                             */
    uint32_t a_blk, b_blk, a2_blk, b2_blk;      /* bytes between consecutive sample tiles; 0 = one constant block  */
    uint16_t a_f0, b_f0, a2_f0, b2_f0;          /* first 1 KiB fragment of the operand inside a block              */
    int32_t mt, nt;                             /* 32-row / 32-column tiles of the product, <= 8 each              */
    float* c;                                   /* fp32 [m][ldc], accumulated with atomics                         */
    float* bias;                                /* fp32 [m] or NULL                                                */
    int32_t ldc, m, n;
    float scale;
    int32_t wg_base, splits;                    /* first workgroup of the job, number of sample-range splits       */
    int32_t n_tiles, pad_;                      /* sample tiles of THIS product's planes (products over other planes in
                                                   the same launch); 0 = the launch's n_sample_tiles                  */
    const int32_t* n_dev;                       /* NULL, or a DEVICE count of the samples this product's planes hold
                                                   (fneus_outside_select): only their tiles are summed                */
} FneusGemmPPJob;

/* one contiguous run of parameters for fneus_adam (all device pointers) */
typedef struct FneusAdamSegment {
    float* param;
    float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    long count;
} FneusAdamSegment;

/* ---- library ------------------------------------------------------------------------------------------------ */
int fneus_version(void);                 /* 100*major + minor */
const char* fneus_last_error(void);      /* host pointer, static storage */

/* Blob geometry, so the host needs no duplicated constants.  which: 0 = SDF network, 1 = colour network (and the two
 * RefColor MLPs), 2 = background NeRF++ (one entry per pack, csrc/fneus_layout.h).
 * out[0] = n_layers, out[1] = total blob bytes, out[2] = extra offset, then per layer 9 ints:
 * fwd_hi, fwd_lo, rev_hi, rev_lo, bias (byte offsets), ksf, ntf, ksr, ntr.  Returns number of ints written. */
int fneus_layout(int which /*0 SDF, 1 colour-shaped, 2 background NeRF, 3 Lvis*/, int32_t* out, int cap);

/* ---- weight packing (every optimiser step) --------------------------------------------------------------- */
/* jobs: device array of PackJob (csrc/fneus_pack.h), maps: device int32 index maps, params: flat fp32 parameters
 * (raw weight_v / weight_g / bias when `rowscale` is given: the weight-norm fold W = g v/||v|| of
 * nn.utils.weight_norm, fields.py:67-68, happens inside the packer), rowscale: per-row g/||v|| or NULL. */
int fneus_pack(const void* jobs, int n_jobs, int n_units, const int32_t* maps, const float* params,
               const float* rowscale, void* blob, fneus_stream_t stream);

/* Weight-norm fold + packing of SEVERAL networks in one launch each (what a training step does at its start: four or five
 * networks, each pair of launches a few microseconds of work).  Task i = the arguments of fneus_rowscale (rows, n_rows,
 * params = raw, rowscale, invnorm; n_rows = 0 for plain Linear networks) and of fneus_pack for one network.  n_tasks <= 8.
 * Replaces the per-module fold of nn.utils.weight_norm (fields.py:67-68, 139-140) for all networks of the step at once. */
typedef struct FneusPackTask {
    const void* jobs; int n_jobs; int n_units; const int* maps; const float* params; float* rowscale; float* invnorm;
    void* blob; const void* rows; int n_rows;
} FneusPackTask;
int fneus_refresh_multi(const FneusPackTask* tasks /*host array*/, int n_tasks, fneus_stream_t stream);

/* rows: device array of RowInfo (csrc/fneus_pack.h), one per weight-normalised output row. */
int fneus_rowscale(const void* rows, int n_rows, const float* raw, float* rowscale, float* invnorm,
                   fneus_stream_t stream);
/* backward of the fold: effective-parameter gradients d_eff (W then b per layer) -> ACCUMULATED into the raw
 * parameter gradients d_raw (weight_v, weight_g, bias); bias_segs: device int4 (src_off, dst_off, count, 0).
 * d_eff is consumed: it is all zeros afterwards (ready for the atomics of the next fneus_dw_gemm_pp). */
int fneus_wn_backward(const void* rows, int n_rows, const void* bias_segs, int n_segs, const float* raw,
                      const float* rowscale, const float* invnorm, float* d_eff, float* d_raw,
                      fneus_stream_t stream);

/* fneus_wn_backward for SEVERAL networks in one launch (task i = its arguments); n_tasks <= 8.  The results are read by the
 * optimiser only, so a single-GPU step can run all of them behind the last weight-gradient GEMM. */
typedef struct FneusWnTask {
    const void* rows; int n_rows; const void* bias_segs; int n_segs; const float* raw; const float* rowscale;
    const float* invnorm; float* d_eff; float* d_raw;
} FneusWnTask;
int fneus_wn_backward_multi(const FneusWnTask* tasks /*host array*/, int n_tasks, fneus_stream_t stream);

/* ---- K1: SDFNetwork.sdf under no_grad  (fields.py:93-95 via renderer.py:199, 430, 515) -------------------- */
int fneus_sdf_fwd(const void* sdf_blob, const float* pts, const float* rays_o, const float* rays_d, const float* t,
                  int m, long n_pts, float* sdf_out /*[n]*/, int prec, fneus_stream_t stream);

/* K1 on the rays marked in ray_mask [n_pts / m] only (rays of m = k x 128 samples, >= 32 768 samples in all): the secondary
 * rays of stage 2 whose primary ray hit the surface (calLvis.py:339-409 marches the hit points' rays only; the fixed-shape step
 * marches 4 per primary ray).  The samples of the other rays get `fill`.  work: int32 [n_pts / 128 + 1] device scratch (the
 * list of 128-sample units to evaluate and its length: no host synchronisation). */
int fneus_sdf_fwd_rays(const void* sdf_blob, const float* rays_o, const float* rays_d, const float* t, int m, long n_pts,
                       const unsigned char* ray_mask, float fill, int32_t* work, float* sdf_out /*[n]*/, int prec,
                       fneus_stream_t stream);

/* ---- K2: SDFNetwork.forward + SDFNetwork.gradient  (fields.py:74-111 via renderer.py:238-242) ------------- */
/* train != 0 additionally writes the a_l / feature planes needed by fneus_sdf_bwd.  Round 6: the feature planes are hi + lo whenever
 * the stash has a feat_lo plane (whatever the gradient precision of the other planes), and feat_out may be NULL for training launches of
 * >= 1024 sample tiles: the consumers of the feature vector on the hot path (fneus_color_fwd with feat = NULL and the planes in its
 * stash's feat_hi / feat_lo; fneus_surface_gather) read the planes -- 1 KiB per sample of fp32 rows neither written nor read.        */
int fneus_sdf_fwd_grad(const void* sdf_blob, const float* pts, const float* rays_o, const float* rays_d,
                       const float* t, int m, long n_pts, const FneusSdfStash* stash /*host struct*/,
                       float* sdf_out /*[n]*/, float* feat_out /*[n][256]*/, float* normal_out /*[n][3]*/, int prec,
                       int train, fneus_stream_t stream);

/* ---- K3: autograd of K2 w.r.t. the SDF weights, incl. the double backward through SDFNetwork.gradient
 *      (create_graph=True, fields.py:104-110).  Consumes the K2 stash, writes the planes in `bufs`; the weight
 *      gradients themselves are produced by fneus_dw_gemm_pp from those planes. */
/* d_feat == NULL (round 6; bf16 gradient planes and >= 1024 sample tiles only, -2 otherwise): the feature rows of the seed are
 * already in bufs->zbar_hi slot 8 as bf16 fragments (fneus_color_bwd with d_feat == NULL, fneus_surface_scatter_plane).        */
int fneus_sdf_bwd(const void* sdf_blob, const float* pts, const float* rays_o, const float* rays_d, const float* t,
                  int m, long n_pts, const FneusSdfStash* stash, const FneusSdfBwdBufs* bufs, const float* d_sdf /*[n]*/,
                  const float* d_feat /*[n][256]*/, const float* d_normal /*[n][3]*/, int prec, fneus_stream_t stream);

/* ---- weight-gradient GEMM (split-K over samples, fp32 atomics into zero-initialised C / bias) ------------------ */
/* Products over fragment planes (FneusGemmPPJob): every weight gradient of the five MLPs.  n_wgs = sum of the jobs'
 * `splits` (at most the CU count: one workgroup per CU); gprec 1 = hi planes only (bf16 operands, fp32 accumulation), 3 = hi + lo planes (hi*hi + hi*lo + lo*hi).
 * Replaces torch autograd's addmm backward for fields.py:86 incl. the double-backward term of fields.py:104-110.       */
int fneus_dw_gemm_pp(const void* jobs_dev /*FneusGemmPPJob[n_jobs] on the device*/, int n_jobs, int n_wgs,
                     long n_sample_tiles, const int32_t* n_samples_dev /*NULL, or a DEVICE count: only the sample tiles of the
                     first *n_samples_dev samples are summed (planes of a launch that took its count from fneus_outside_select)*/,
                     int gprec, fneus_stream_t stream);
/* The same products with BIT-REPRODUCIBLE results (FNEUS_DETERMINISTIC=1 in the Python layer): every workgroup writes its
 * partial tile to `scratch` (n_wgs x (256 x 256 + 256) floats, caller-owned), a second launch adds the partials of a product
 * in split order.  The reference's addmm backward (exp_runner.py:179-181 via autograd) is deterministic on CPU; the default
 * entry point above sums split-K partials with fp32 atomics in arrival order. */
int fneus_dw_gemm_pp_det(const void* jobs_dev, int n_jobs, int n_wgs, long n_sample_tiles, const int32_t* n_samples_dev,
                         int gprec, float* scratch, long scratch_floats, fneus_stream_t stream);

/* ---- K4: RenderingNetwork.forward, mode 'idr'  (fields.py:150-175 via renderer.py:278) ---------------------- */
/* view directions: `dirs` [n][3], or NULL -> rays_d[n/m].  train != 0 writes the stash planes for the backward.   */
int fneus_color_fwd(const void* col_blob, const float* pts, const float* rays_o, const float* rays_d, const float* t,
                    int m, long n_pts, const float* dirs, const float* normal /*[n][3]*/, const float* feat /*[n][256]*/,
                    const FneusColStash* stash /*host struct, may be NULL when !train*/, float* rgb_out /*[n][3]*/,
                    int prec, int train, fneus_stream_t stream);

/* autograd of the above: d_rgb -> d_feat [n][256], d_normal [n][3]; writes zbar planes (weight-gradient operands).
 * d_feat == NULL (with stash->dfeat_hi set; bf16 gradient planes and a launch of >= 1024 sample tiles only, -2 otherwise): the
 * feature cotangent leaves the launch as the bf16 fragments the SDF network's backward and weight-gradient product read anyway
 * (fields.py:150-175 through autograd: the same values, rounded where fneus_sdf_bwd would round them).                     */
int fneus_color_bwd(const void* col_blob, long n_pts, const float* d_rgb, const float* rgb, const FneusColStash* stash,
                    float* d_feat, float* d_normal, int prec, fneus_stream_t stream);

/* Weight and bias gradient of the colour network's OUTPUT layer (lin4, 256 -> 3: fields.py:170-174 through autograd's addmm backward)
 * with exact operands -- gradient precision 2, the default of the training step:  dW[c][k] += sum_n zout[n][c] u3[n][k],
 * db[c] += sum_n zout[n][c],  zout = d_rgb * rgb * (1 - rgb) formed in fp32.  u3_hi / u3_lo: slot 3 of FneusColStash.u (fragment
 * planes [tiles][16][64][8] bf16; u3_lo may be NULL: hi plane alone).  dW [3][256] and db [3] (or NULL) are ACCUMULATED into
 * (fp32 atomics).  It is the one product of a step whose bf16 operand rounding exceeds the exact mode's gradient bounds.          */
/* scratch: fneus_color_out_dw_scratch_floats() floats, ZERO at the first call; the call leaves it zero (replicas of the sums: an
 * address receives 16 atomic adds, not one per workgroup; a second, one-workgroup launch folds them into dW / db).                */
/* fold_src / fold_n / fold_dst (or NULL / 0 / NULL): the one-workgroup fold launch ALSO adds sum(fold_src[0 .. fold_n)) to *fold_dst --
 * the step's other small reduction, the variance parameter's gradient (per-ray d inv_s of fneus_composite_bwd), rides along instead
 * of costing a reduction launch and an accumulation launch of its own.                                                           */
int fneus_color_out_dw(const void* u3_hi, const void* u3_lo, const float* d_rgb /*[n][3]*/, const float* rgb /*[n][3]*/, long n_pts,
                       float* dW, float* db, float* scratch, const float* fold_src, int fold_n, float* fold_dst, fneus_stream_t stream);
int fneus_color_out_dw_scratch_floats(void);

/* ---- K4': RefColor.forward, the surface colour head  (fields.py:271-335 via renderer.py:330-339) ------------- */
/* Its two MLPs have the colour network's shape and run on the same kernels.  head 1 = net_cd: [pts | PE4(n) | feature]
 * -> diffuse rgb out[n][3];  head 2 = viewdir_mlp + net_cs: [n | pts | PE4(reflect(-d, n/|n|)) | feature] -> specular
 * out[n][0] (columns 1, 2 are padding).  Plain nn.Linear layers (no weight norm); blobs come from fneus_pack.  The
 * sRGB transfer, clipping and the two-sample blend of renderer.py:336-339 stay with the caller.                       */
int fneus_refcolor_fwd(const void* blob, int head, const float* pts, const float* rays_o, const float* rays_d,
                       const float* t, int m, long n_pts, const float* dirs, const float* normal /*[n][3]*/,
                       const float* feat /*[n][256]*/, const FneusColStash* stash, float* out /*[n][3]*/, int prec,
                       int train, fneus_stream_t stream);

/* autograd of the above: d_out [n][3] (head 2: columns 1, 2 must be zero) -> d_feat [n][256], d_normal [n][3] (through
 * PE4(n), the reflection and the normalisation); view directions as in the forward call (dirs, or rays_d[n/m]).      */
int fneus_refcolor_bwd(const void* blob, int head, long n_pts, const float* rays_d, int m, const float* dirs,
                       const float* normal, const float* d_out, const float* out, const FneusColStash* stash,
                       float* d_feat, float* d_normal, int prec, fneus_stream_t stream);

/* Both heads in ONE launch (the training step uses these: at 2 samples per ray a head is 32 workgroups, so a launch
 * lasts one tile's chain whatever it contains).  spec_out [n][3] (column 0); d_feat2 [2][n][256] and d_normal2 [2][n][3]
 * hold the diffuse head's input gradients in slice 0 and the specular head's in slice 1 (the caller adds them).          */
int fneus_refcolor_fwd_both(const void* blob_cd, const void* blob_vd, const float* pts, const float* rays_o,
                            const float* rays_d, const float* t, int m, long n_pts, const float* dirs, const float* normal,
                            const float* feat, const FneusColStash* stash_cd, const FneusColStash* stash_vd,
                            float* diffuse_out, float* spec_out, int prec, int train, fneus_stream_t stream);
int fneus_refcolor_bwd_both(const void* blob_cd, const void* blob_vd, long n_pts, const float* rays_d, int m,
                            const float* dirs, const float* normal, const float* d_diffuse, const float* d_spec,
                            const float* diffuse, const float* spec, const FneusColStash* stash_cd,
                            const FneusColStash* stash_vd, float* d_feat2, float* d_normal2, int prec,
                            fneus_stream_t stream);

/* ---- per-ray tail of the training step ------------------------------------------------------------------------ */
/* Optional argument of fneus_surface_gather / fneus_stage1_loss (NULL = none): device ranges (e.g. the forward / reverse weight
 * fragments of the RefColor MLPs) that extra workgroups of THIS launch read into the L2 caches for the launch that follows it.
 * A host struct read during the call only: the library keeps no pointer of it, the ranges must be live while the launch (or a
 * graph that captured it) runs -- the caller that owns the buffers names them per call.                                      */
#define FNEUS_MAX_WARM_RANGES 12
typedef struct FneusWarmRanges {
    int n;                                      /* ranges in use (<= FNEUS_MAX_WARM_RANGES) */
    const void* ptr[FNEUS_MAX_WARM_RANGES];     /* device pointers */
    long bytes[FNEUS_MAX_WARM_RANGES];
} FneusWarmRanges;

/* The two samples bracketing the first SDF sign change of every ray (renderer.py:290-293, 316-327), packed for the
 * RefColor heads: sel [2B] (row index into the B*n samples), t_sel [2B], feat_sel [2B][256], normal_sel [2B][3].
 * Rays without a sign change (sdf_mask 0) select samples 0 and 1, as the reference's dense formulation does.           */
int fneus_surface_gather(const int32_t* min_idx, const unsigned char* sdf_mask, const float* mid_z /*[B][n]*/,
                         const float* feat /*[B*n][256], or NULL: the rows are read from the planes*/,
                         const void* feat_hi, const void* feat_lo /*FneusSdfStash.feat_hi / feat_lo (round 6); lo or both may be NULL*/,
                         const float* normal /*[B*n][3]*/, int n_rays, int n,
                         int32_t* sel, float* t_sel, float* feat_sel, float* normal_sel, const FneusWarmRanges* warm /*or NULL*/,
                         fneus_stream_t stream);

/* The way back of fneus_surface_gather: d_feat[sel[i]][:] += sum_h d_feat_heads[h][i][:], d_normal likewise (the gradients of the
 * RefColor heads with respect to the gathered rows, renderer.py:316-327 through autograd's index backward).  The selected rows are
 * distinct.  Either heads pointer may be NULL.  Replaces two sums over the heads and two index_add_ launches.                    */
int fneus_surface_scatter(const int32_t* sel /*[R]*/, const float* d_feat_heads /*[H][R][256]*/, const float* d_normal_heads /*[H][R][3]*/,
                          int n_heads, long n_rows, float* d_feat /*[N][256]*/, float* d_normal /*[N][3]*/, fneus_stream_t stream);
/* The same with the feature cotangent held as bf16 FRAGMENTS (FneusColStash.dfeat_hi = slot 8 of FneusSdfBwdBufs.zbar_hi,
 * [tiles][16][512], csrc/fneus_pp.h) instead of fp32 rows: the selected rows' elements are read, summed with the heads' in fp32 and
 * rounded back (round 6: the feature cotangent never exists as rows in the training step).  d_normal stays rows.                */
int fneus_surface_scatter_plane(const int32_t* sel /*[R]*/, const float* d_feat_heads /*[H][R][256]*/, const float* d_normal_heads /*[H][R][3]*/,
                                int n_heads, long n_rows, void* dfeat_hi /*[tiles][16][512] bf16*/, long n_pts, float* d_normal /*[N][3]*/,
                                fneus_stream_t stream);

/* RefColor shading (linear->sRGB, clip: fields.py:262-268, 331-335), the two-sample blend (renderer.py:336-343), the
 * training losses (exp_runner.py:141-177: colour L1, surface L1, eikonal, mask BCE) and the gradient of the total loss
 * with respect to every differentiable input, in one launch.  diffuse / spec are the outputs of fneus_refcolor_fwd
 * heads 1 / 2 on the gathered samples.  losses[9] = total, colour, surface, eikonal, mask, psnr, mask_sum, mask_sdf_sum, and the
 * total once more (round 6: the slot a caller hands on as "the loss", a tensor of its own, without a copy launch).           */
int fneus_stage1_loss(const float* color /*[B][3]*/, const float* true_rgb /*[B][3]*/, const float* mask_in /*[B]*/,
                      const float* wsum /*[B]*/, const float* eik_num /*[B]*/, const float* eik_den /*[B]*/,
                      const float* diffuse /*[2B][3]*/, const float* spec /*[2B][3], column 0*/, const float* wpair /*[B][2]*/,
                      const unsigned char* sdf_mask /*[B]*/, const float* norms /*[4] or NULL*/, int n_rays,
                      float igr_weight, float mask_weight, float surface_weight, float* losses, float* surface_color /*[B][3]*/, float* specular_color,
                      float* diffuse_color, float* d_color, float* d_wsum, float* d_eiknum, float* d_wpair,
                      float* d_diffuse /*[2B][3]*/, float* d_spec /*[2B][3]*/, const FneusWarmRanges* warm /*or NULL*/,
                      fneus_stream_t stream);

/* Data parallel: norms[4] = (sum mask, sum mask*sdf_mask, sum eik_den, ray count) of this rank's batch.  Sum them over
 * the ranks (a 4-float all-reduce) and pass the result to fneus_stage1_loss: its loss terms and gradients are then this
 * rank's share of the GLOBAL batch's, and the gradient all-reduce is a plain sum.  norms = NULL: one batch, one rank. */
int fneus_stage1_norms(const float* mask_in, const unsigned char* sdf_mask, const float* eik_den, int n_rays,
                       float mask_weight, float* norms, fneus_stream_t stream);

/* (ray_mask / fill / work as in fneus_sdf_fwd_rays; ray_mask NULL: every sample) */

/* ---- K6: hierarchical sampler pieces (one wavefront per ray, 2 <= samples per ray <= 256; fneus_upsample: <= 512) -- */
/* NeuSRenderer.up_sample + sample_pdf(det=True)  (renderer.py:152-189, 43-77): z [B][m], sdf [B][m] -> z_new [B][k]  */
int fneus_upsample(const float* rays_o, const float* rays_d, const float* z, const float* sdf, int n_rays, int m, int k,
                   float inv_s, float* z_new, fneus_stream_t stream);
/* (m <= 512 since round 2: the stage-2 secondary rays up-sample 512 coarse samples, calLvis.py:55-90, same algorithm) */
/* cat_z_vals (renderer.py:191-205): stable sort-merge of (z_old | z_new); s_old/s_new/s_out may be NULL (last step) */
int fneus_merge(const float* z_old, const float* s_old, int m, const float* z_new, const float* s_new, int k, int n_rays,
                float* z_out, float* s_out, fneus_stream_t stream);
/* cat_z_vals of one up-sampling step fused with up_sample of the next (renderer.py:433-446 is a per-ray recurrence):
 * z_out, s_out [B][m + k] = merge; z_next [B][k_next] = up_sample(z_out, s_out, inv_s); z_final (may be NULL; the last
 * step, renderer.py:445 last=True) [B][m + k + k_next] = merge(z_out | z_next).  Bit-identical to the separate calls.
 * z_old must be ASCENDING per ray (it is: fneus_ray_setup's depths or the z_out of the previous step) -- the merge ranks an entry
 * by its position in its own run plus a search of the other run (round 5); z_new may be in any order.
 * dists, mid_z (may be NULL; with z_final): fneus_sections of z_final in the same launch -- what render_core asks for next. */
int fneus_merge_upsample(const float* rays_o, const float* rays_d, const float* z_old, const float* s_old, int m,
                         const float* z_new, const float* s_new, int k, int n_rays, float inv_s, int k_next, float* z_out,
                         float* s_out, float* z_next, float* z_final, float sample_dist, float* dists, float* mid_z,
                         fneus_stream_t stream);

/* K1 on the k new depths of an up-sampling step (ray form, k = 16 or 32) AND that step's fneus_merge_upsample in ONE launch (round 6): the
 * sampler of renderer.py:430-446 is a per-ray recurrence with one SDF evaluation per step, a 32-sample tile of the evaluation is 32 / k
 * whole rays, and the workgroup that evaluated a tile runs the merge of its rays behind it (new sdf values from LDS) -- same values as
 * fneus_sdf_fwd + fneus_merge_upsample, bit for bit.  s_new_out [B][k] or NULL.  Returns -3 for a shape this form does not take
 * (k other than 16 / 32, >= 1024 tiles, more than 256 merged depths): the caller then runs the two launches.                       */
int fneus_sdf_fwd_merge_upsample(const void* sdf_blob, const float* rays_o, const float* rays_d, const float* z_old, const float* s_old,
                                 int m, const float* z_new, int k, int n_rays, float inv_s, int k_next, float* z_out, float* s_out,
                                 float* z_next, float* z_final, float sample_dist, float* dists, float* mid_z, float* s_new_out, int prec,
                                 fneus_stream_t stream);

/* ALL remaining steps of the hierarchical sampler in one launch (renderer.py:433-446, steps i = 1 .. up_sample_steps - 1; round 6): a
 * workgroup keeps its 32 / k rays through every step -- it evaluates the depths its own merge has just drawn -- so the per-ray recurrence
 * costs one launch instead of one per step.  The caller chains the buffers: z_old[j + 1] = z_out[j], s_old[j + 1] = s_out[j],
 * z_new[j + 1] = z_next[j]; the last step writes z_final [B][m_last + k + k_next] and, with mid_z given, its sections.  Bit-identical
 * to n_steps calls of fneus_sdf_fwd_merge_upsample.  -3: a shape this form does not take (more than 4 steps, k not 16 / 32, >= 1024
 * sample tiles).                                                                                                                 */
typedef struct FneusSamplerStep {
    const float *z_old, *s_old;   /* [B][m]      depths and sdf values merged so far                       */
    int m;
    const float* z_new;           /* [B][k]      the depths this step evaluates                            */
    float inv_s;                  /*             64 * 2^i of the step                                      */
    float *z_out, *s_out;         /* [B][m + k]  the merged arrays                                         */
    float* z_next;                /* [B][k_next] the next step's (or, in the last step, the final) new depths */
} FneusSamplerStep;
int fneus_sdf_fwd_merge_upsample_steps(const void* sdf_blob, const float* rays_o, const float* rays_d, int n_steps,
                                       const FneusSamplerStep* steps /*host array*/, int k, int n_rays, int k_next, float* z_final,
                                       float sample_dist, float* dists, float* mid_z, int prec, fneus_stream_t stream);
/* One training batch [B][10] = rays_o, rays_d, rgb, mask per row (what Dataset.gen_random_rays_at returns, dataset.py:133-151)
 * -> the four contiguous arrays the kernels take (exp_runner.py:134-139 slices the same columns). */
int fneus_split_batch(const float* data /*[B][10]*/, int n_rays, float* rays_o /*[B][3]*/, float* rays_d /*[B][3]*/,
                      float* rgb /*[B][3]*/, float* mask /*[B]*/, fneus_stream_t stream);

/* Coarse depths: z_vals [B][n_samples] = near + (far - near) * linspace(0, 1, n_samples) (renderer.py:393-395), plus the
 * per-ray jitter (t_rand[b] - 0.5) * 2 / n_samples when t_rand [B] (uniform in [0,1)) is given (renderer.py:405-409).
 * near / far [B] as passed to NeuSRenderer.render, or both NULL: the unit-sphere bounds of dataset.py:186-192
 * (near_far_from_sphere) are computed from the rays.                                                                 */
int fneus_ray_setup(const float* rays_o, const float* rays_d, const float* near, const float* far, const float* t_rand,
                    int n_rays, int n_samples, float* z_vals, fneus_stream_t stream);

/* section lengths and mid points of render_core (renderer.py:223-226) */
int fneus_sections(const float* z, int n_rays, int n, float sample_dist, float* dists, float* mid_z, fneus_stream_t stream);

/* Embedder.embed (embedder.py:23-36) of constant inputs (points, directions): x [n][d] -> out [n][d (1 + 2 n_freqs)] =
 * [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]; one launch instead of 2 L + 2 element-wise kernels.
 * (The fused MLP kernels encode their inputs themselves; this serves the torch-side networks of stages 2 and 3.)           */
int fneus_embed(const float* x, long n_rows, int d, int n_freqs, float* out, fneus_stream_t stream);

/* ---- Ray generation (models/dataset.py:115-151): pixel -> K^-1 (x, y, 1) -> normalise -> R v; o = pose[:3, 3] ------------- */
/* Dataset.gen_random_rays_at (dataset.py:133-151): the rays, colours and mask values of n integer pixels of ONE image.
 * intrinsics_inv, pose: that image's [4][4] matrices (row major); image, mask: its float [H][W][3] planes (BGR / 256 as the
 * reference's cv2 loader gives them, dataset.py:61-67); pixels_x / pixels_y: int64 [n] (torch.randint);
 * out [n][10] = rays_o, rays_d, rgb, mask[..., :1].  Everything is device resident: no host copy per training step.   */
int fneus_gen_random_rays(const float* intrinsics_inv, const float* pose, const float* image, const float* mask, int H, int W,
                          const long long* pixels_x, const long long* pixels_y, int n_rays, float* out, fneus_stream_t stream);
/* Dataset.gen_rays_at (dataset.py:115-131): all rays of one camera at pixel positions tx [nx] x ty [ny] (the caller's
 * torch.linspace(0, W-1, W // l), linspace(0, H-1, H // l)) -> rays_o, rays_v [ny][nx][3] (image-row major, as returned
 * by the reference after its transpose).                                                                              */
int fneus_gen_rays_grid(const float* intrinsics_inv, const float* pose, const float* tx, const float* ty, int nx, int ny,
                        float* rays_o, float* rays_v, fneus_stream_t stream);

/* ---- Stage 2 (lvis.py): NeuSRenderer.lvis_render (renderer.py:567-627) and cal_indiLgt (calLvis.py:339-409) -------------- */
/* First surface hit of every ray: idx = first sample with sign(sdf) = -1; sdf_mask = idx exists & idx >= 1 & the ray has a
 * sample inside the unit sphere (inside_mask [B] if given, else computed from the points); z_surf = zero crossing of the SDF
 * between samples idx-1 and idx by linear interpolation; pts_surf = o + d z_surf   (renderer.py:586-604 = calLvis.py:178-196;
 * rays without a hit get z_surf = 0).  With normal [B][n][3] and dists [B][n] also compute_weight (calLvis.py:93-150, NeuS
 * alpha at cos_anneal_ratio 0): occlusion [B] = sum of the weights of the samples inside the unit sphere, weights [B][n]
 * (may be NULL).  2 <= n <= 256.                                                                                          */
int fneus_ray_hit(const float* rays_o, const float* rays_d, const float* mid_z, const float* sdf, const float* dists /*may be NULL*/,
                  const float* normal /*may be NULL*/, const unsigned char* inside_mask /*may be NULL*/, int n_rays, int n,
                  float inv_s, unsigned char* sdf_mask, float* z_surf, float* pts_surf, float* occlusion /*NULL without normal*/,
                  float* weights /*may be NULL*/, fneus_stream_t stream);
/* sample_dirs (calLvis.py:302-320) on the draws of calLvis.py:351-355: n_dirs directions per surface point,
 * theta = 2 pi u_theta, phi = asin(0.95 u_z); origins [n_pts*n_dirs][3] = the surface point of each direction (:357).      */
int fneus_sample_dirs(const float* surf /*[n_pts][3]*/, const float* normal /*[n_pts][3]*/, const float* u_theta /*[n_pts][n_dirs]*/,
                      const float* u_z /*[n_pts][n_dirs]*/, int n_pts, int n_dirs, float* origins, float* dirs,
                      fneus_stream_t stream);

/* ---- Stage 3 (mateIllu.py): per-lobe light visibility, get_diffuse_visibility (inverRender.py:128-192) ------------------------ */
/* IndirectLight's output transform (models/fields.py:395-413), no gradient: raw [N][6] -> lgtSGs [N][7] = (cos th sin ph, sin th sin ph,
 * cos ph, 30 sigmoid(o2) + 0.1, relu(o3..5)), N = points x lobes. */
int fneus_indir_sgs(const float* raw, long n_lobes_total, float* sgs, fneus_stream_t stream);

/* The direction set of get_diffuse_visibility (inverRender.py:133-161): lobes [M][3], lambdas [M] (sharpness), u_theta / u_phi
 * [M][S] uniform draws -> dirs [M][S][3] inside each lobe's cone, weights [M][S] = exp(lambda (dir . axis - 1)). */
int fneus_vis_sample_dirs(const float* lobes, const float* lambdas, const float* u_theta, const float* u_phi, int n_lobes, int n_samp,
                          float* dirs, float* weights, fneus_stream_t stream);
/* the same from the light-SG table lgtSGs [M][7] of EnvmapMaterialNetwork: lobes = sg[0..2] / (|sg[0..2]| + 1e-6), lambdas = |sg[3]|
 * (render_with_all_sg, inverRender.py:420-421) taken inside the launch. */
int fneus_vis_sample_dirs_sgs(const float* lgt_sgs, const float* u_theta, const float* u_phi, int n_lobes, int n_samp, float* dirs,
                              float* weights, fneus_stream_t stream);
/* The inputs of EnvmapMaterialNetwork's MLPs (inverRender.py:530-545), no gradient: points, ray_dirs, normals [n][3] ->
 * n_unit = normal / (|normal| + 1e-6), view_dirs = -ray_dir / (|ray_dir| + 1e-6) [n][3], enc_pts [n][63] = embed(point, 10) (the BRDF
 * encoder's input; embedder.py:23-36), x_cs [n][90] = [embed(point, 10) | embed(2 (v . n) n - v, 4)] (net_cs's input). */
int fneus_material_inputs(const float* points, const float* ray_dirs, const float* normals, int n, float* n_unit, float* view_dirs,
                          float* enc_pts, float* x_cs, fneus_stream_t stream);

/* ---- sRGB transfer curves (models/math_utils.py:138-152; RefColor, fields.py:329-335; stage-3 tone mapping, inverRender.py:13-18)
 * as one element-wise launch and one for the adjoint.  mode bit 0: 0 = linear -> sRGB, 1 = sRGB -> linear; bit 1: clip to [0, 1]
 * behind the curve (zero gradient outside). */
int fneus_srgb_fwd(const float* x, long n, int mode, float* y, fneus_stream_t stream);
int fneus_srgb_bwd(const float* x, const float* dy, long n, int mode, float* dx, fneus_stream_t stream);

/* ---- stage 2: predicted indirect radiance from the RAW output of the IndirectLight MLP (models/fields.py:395-413 output
 * transform + models/calLvis.py:323-336 query_indir_illum) and its adjoint ------------------------------------------------- */
/* raw [n][L][6] (theta, phi, sharpness, amplitude rgb before their sigmoid / relu), dirs [n][S][3] unit directions (no
 * gradient) -> radiance [n][S][3] = sum_l mu_l exp(lambda_l (axis_l . d_s - 1)).  L <= 64, S <= 8. */
int fneus_indir_illum_fwd(const float* raw, const float* dirs, int n, int n_lobes, int n_dirs, float* radiance,
                          fneus_stream_t stream);
int fneus_indir_illum_bwd(const float* raw, const float* dirs, const float* d_radiance /*[n][S][3]*/, int n, int n_lobes,
                          int n_dirs, float* d_raw /*[n][L][6]*/, fneus_stream_t stream);

/* ---- stages 2 / 3: the TRAINED plain MLPs on a few hundred rows -- Lvis and IndirectLight (models/fields.py:338-413: nn.Linear +
 * ReLU, Lvis ends in a sigmoid), the BRDF auto-encoder and net_cs of EnvmapMaterialNetwork (models/inverRender.py:451-598:
 * nn.Linear + LeakyReLU(0.2), net_cs ends in a sigmoid).  Replaces torch's per-Linear GEMM + activation + bias-gradient launches:
 * one launch per layer and direction, one for ALL weight / bias gradients of the listed layers.  fp32 products and sums
 * (fp32 MFMA), bit-reproducible.  Every array row-major fp32; the jobs of one call are independent of each other (a call = one
 * launch of up to 16 jobs: the same layer index of several networks, or all layers of a network for the parameter gradients).
 * act / act_in: 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 sigmoid. */
typedef struct FneusMlpJob {
    const float* x;        /* [rows][n_in] the layer's input (the previous layer's output) */
    const float* weight;   /* [n_out][n_in] nn.Linear.weight */
    const float* bias;     /* [n_out] or NULL */
    float* y;              /* [rows][n_out] the layer's output act(x W^T + b) */
    const float* dy;       /* backward: [rows][n_out] gradient of y if act != 0 (act'(y) is applied while it is loaded), of the
                              pre-activation if act == 0 (what fneus_mlp_backward_input of the layer above has written) */
    float* dx;             /* backward_input: [rows][n_in] gradient of the PRE-activation of the layer below (act_in' from x
                              applied); with act_in == 0 the gradient of x itself */
    float* d_weight;       /* backward_params: [n_out][n_in], OVERWRITTEN */
    float* d_bias;         /* backward_params: [n_out], OVERWRITTEN; NULL = not wanted */
    int rows, n_in, n_out;
    int act;               /* this layer's activation (see dy) */
    int act_in;            /* the activation that produced x (backward_input) */
} FneusMlpJob;
int fneus_mlp_forward(const FneusMlpJob* layers /*host array*/, int n_layers, fneus_stream_t stream);          /* x, weight, bias -> y */
int fneus_mlp_backward_input(const FneusMlpJob* layers /*host array*/, int n_layers, fneus_stream_t stream);   /* dy (y), weight, x -> dx */
int fneus_mlp_backward_params(const FneusMlpJob* layers /*host array*/, int n_layers, fneus_stream_t stream);  /* dy (y), x -> d_weight, d_bias */

/* lvis_blob: fneus_pack output for layout 3 (the Lvis network, fields.py:338-369, plain Linear layers).  points, normals
 * [n_pts][3] (unit normals); dirs [n_lobes][32][3]: the sampled directions around every light lobe (inverRender.py:158-161);
 * weights [n_lobes][32] = exp(lambda (d . axis - 1)) (:186).  vis [n_lobes][n_pts] = sum_s [n . d_s > 1e-6] Lvis(p, d_s) w_s /
 * (sum_s w_s + 1e-6)  (:169-188).  One 32-sample MFMA tile per (point, lobe) pair: n_dirs must be 32.                      */
int fneus_lvis_visibility(const void* lvis_blob, const float* points, const float* normals, const float* dirs,
                          const float* weights, const unsigned char* point_mask /*[n_pts] or NULL: 0 = skip the point (vis = 0)*/,
                          int n_pts, int n_lobes, int n_dirs, float* vis, int prec, fneus_stream_t stream);
/* prec 2 (round 5; this entry point only): ONE fp16 product per multiplication -- a lobe's visibility averages up to 32 sigmoid
 * outputs and stays within 3e-5 of the parity mode's at a third of its matrix work; lvis_blob is then the output of
 * fneus_lvis_h16_pack (the packed blob with every forward weight as one fp16 value; fneus_lvis_blob_bytes() bytes). */
size_t fneus_lvis_blob_bytes(void);
int fneus_lvis_h16_pack(const void* lvis_blob, void* out, fneus_stream_t stream);

/* Spherical-Gaussian rendering of stage 3: render_with_sg (inverRender.py:314-449) with lambda_trick (:83-103), hemisphere_int
 * (:106-125) and integrate_rgb (:264-283) for the n_direct light SGs lgt_sgs [n_direct][7] (with per-lobe visibility vis
 * [n_direct][n_pts]) and the n_indirect SGs indir_sgs [n_pts][n_indirect][7] of every point (no visibility).  normal, view
 * [n_pts][3] unit vectors; material [n_pts][7] = roughness, diffuse albedo[3], specular albedo[3].
 * out [n_pts][4][3] = the lobe sums BEFORE integrate_rgb's clamp: direct specular, direct diffuse, indirect specular, indirect
 * diffuse.  The backward returns d_material [n_pts][7] and ACCUMULATES (atomics) into d_lgt_sgs [n_direct][7]; the indirect
 * SGs, normals, view directions and visibilities are constants (frozen networks, mateIllu.py:83-95).                        */
int fneus_sg_render_fwd(const float* lgt_sgs, const float* indir_sgs, const float* vis, const float* normal, const float* view,
                        const float* material, int n_pts, int n_direct, int n_indirect, float specular_reflectance, float* out,
                        fneus_stream_t stream);
int fneus_sg_render_bwd(const float* lgt_sgs, const float* indir_sgs, const float* vis, const float* normal, const float* view,
                        const float* material, int n_pts, int n_direct, int n_indirect, float specular_reflectance,
                        const float* d_out, float* d_material, float* d_lgt_sgs, fneus_stream_t stream);
/* The same with the material taken straight from the outputs of EnvmapMaterialNetwork's two heads: brdf [n_pts][4] = (diffuse albedo
 * rgb, raw roughness) -- the sigmoid of the BRDF decoder -- and cs [n_pts] (net_cs): roughness = 0.9 raw + 0.09 (inverRender.py:557),
 * specular albedo = cs in all three channels (:560).  The backward returns d_brdf [n_pts][4] and d_cs [n_pts]. */
int fneus_sg_render_heads_fwd(const float* lgt_sgs, const float* indir_sgs, const float* vis, const float* normal, const float* view,
                              const float* brdf, const float* cs, int n_pts, int n_direct, int n_indirect, float specular_reflectance,
                              float* out, fneus_stream_t stream);
int fneus_sg_render_heads_bwd(const float* lgt_sgs, const float* indir_sgs, const float* vis, const float* normal, const float* view,
                              const float* brdf, const float* cs, int n_pts, int n_direct, int n_indirect, float specular_reflectance,
                              const float* d_out, float* d_brdf, float* d_cs, float* d_lgt_sgs, fneus_stream_t stream);
/* The two L1 terms of a stage-2 step (lvis.py:164-170) over the n primary rays, 4 secondary rays each: out[0] = sum over rays
 * with a hit of |gt_lvis - pre_lvis| / (4 n_hit + 1e-6), out[1] = the same of the traced radiance [n][4][3] / (12 n_hit + 1e-6),
 * out[2] = n_hit; d_pre_lvis [n][4], d_pre_rad [n][4][3]: gradients of out[0] / out[1] (zero rows without a hit). */
int fneus_stage2_loss(const float* gt_lvis, const float* pre_lvis, const float* gt_rad, const float* pre_rad,
                      const unsigned char* hit, int n, float* out /*[3]*/, float* d_pre_lvis, float* d_pre_rad, fneus_stream_t stream);
/* The image terms of a stage-3 step (mateIllu.py:152-172) over n rays: w = mask x hit; out[0] = sum |(rgb - true_rgb) w| /
 * (sum w + 1e-5), out[1] = psnr, out[2] = sum w; d_rgb [n][3] = d out[0] / d rgb. */
int fneus_stage3_loss(const float* rgb, const float* true_rgb, const float* mask /*[n]*/, const unsigned char* hit /*[n]*/, int n,
                      float* out /*[3]*/, float* d_rgb, fneus_stream_t stream);
/* The latent-sparsity term of stage 3 (inverRender.py:609-612): latent [n][32], point_mask [n] (NULL = every point), rho in
 * (0, 1) -> stats [34] = rho_hat [32] (mean sigmoid over the marked points), their number, kl (0 without a marked point); and the
 * adjoint d_latent [n][32] for the cotangent d_kl [1] (device scalar). */
/* activated != 0: `latent` holds sigmoid(latent) already (fneus_mlp_forward applied it as the encoder's last activation); the adjoint is
 * then the gradient of that activated tensor. */
int fneus_latent_kl_fwd(const float* latent, const unsigned char* point_mask, int n, float rho, int activated, float* stats,
                        fneus_stream_t stream);
int fneus_latent_kl_bwd(const float* latent, const unsigned char* point_mask, int n, float rho, int activated, const float* stats,
                        const float* d_kl, float* d_latent, fneus_stream_t stream);
/* The colour a stage-3 training step reads, from the four lobe sums of fneus_sg_render_fwd [n][4][3]: clamp each to [0, 1],
 * env = clamp(direct specular + diffuse), indir = clamp(indirect specular + diffuse) (0 when has_indir == 0), rgb =
 * clip(linear -> sRGB (env + indir)) -- inverRender.py:277, 440, 306-309 -- and the adjoint (torch.clamp / torch.clip pass the
 * gradient on the closed interval). */
int fneus_sg_combine_fwd(const float* sums, long n, int has_indir, float* rgb /*[n][3]*/, fneus_stream_t stream);
int fneus_sg_combine_bwd(const float* sums, const float* d_rgb, long n, int has_indir, float* d_sums /*[n][4][3]*/,
                         fneus_stream_t stream);

/* ---- K7: background NeRF++ of the womask configurations  (fields.py:233-259 NeRF.forward via renderer.py:112-149) ---- */
/* pts4 [n][4] = (p/|p|, 1/|p|) of the background samples, dirs [n][3]; outputs are RAW: density [n] (alpha_linear) and
 * rgb [n][3] (rgb_linear) -- softplus, sigmoid and the compositing of renderer.py:139-149 stay with the caller.
 * nerf_blob: fneus_pack output for layout 2.  train != 0 writes the stash planes for the backward.                     */
int fneus_nerf_bg_fwd(const void* nerf_blob, const float* pts4, const float* dirs, long n_pts,
                      const FneusNerfStash* stash /*host struct, may be NULL when !train*/, float* density /*[n]*/,
                      float* rgb /*[n][3]*/, int prec, int train,
                      const int32_t* n_dev /*NULL, or a DEVICE count <= n_pts: only the first *n_dev rows are evaluated (the list
                      of fneus_outside_select); buffers and stash planes keep the shape of n_pts*/, fneus_stream_t stream);

/* autograd of the above w.r.t. the parameters (the inputs are constants: every z is sampled under no_grad): writes the
 * dL/dz planes of the stash; the weight / bias gradients are then ONE fneus_dw_gemm_pp launch over stash planes.          */
int fneus_nerf_bg_bwd(const void* nerf_blob, long n_pts, const float* d_density /*[n]*/, const float* d_rgb /*[n][3]*/,
                      const FneusNerfStash* stash, int prec, const int32_t* n_dev /*as in fneus_nerf_bg_fwd*/,
                      fneus_stream_t stream);

/* z_vals_outside of NeuSRenderer.render (renderer.py:397-400, 411-419) -> z [B][n_out]: linspace(1e-3, 1 - 1/(n_out+1), n_out),
 * jittered inside its cells by u [B][n_out] (NULL: no jitter), flipped, far / t + 1 / n_samples; far [B] or NULL = the rays'
 * unit-sphere bound (dataset.py:186-192). */
int fneus_outside_z(const float* rays_o, const float* rays_d, const float* far, const float* u, int n_rays, int n_out,
                    int n_samples, float* z, fneus_stream_t stream);

/* ---- the element-wise work of render_core_outside around K7 (renderer.py:112-149) ------------------------------------ */
/* z [B][nt]: the merged inside + outside depths (renderer.py:453).  -> dists [B][nt] (last section = sample_dist), pts4
 * [B*nt][4] = (p / |p|, 1 / |p|) at the section mid points with |p| clipped to [1, 1e10], dirs [B*nt][3] = the ray direction. */
int fneus_outside_points(const float* rays_o, const float* rays_d, const float* z, int n_rays, int nt, float sample_dist,
                         float* pts4, float* dirs, float* dists, fneus_stream_t stream);
/* The background samples whose value render_core USES (renderer.py:350-356 blends sample i < n as x_i inside_i + bg_i (1 - inside_i):
 * inside the unit sphere the background value is multiplied by exactly 0, forward and backward).  z_core [B][n]: the depths
 * render_core gets; z_feed [B][nt]: the merged depths the background is evaluated at (renderer.py:453), nt <= 256.  Listed, in
 * ray-major order: every sample i >= n and every i < n whose section mid point (renderer.py:228-231) is not inside the unit sphere.
 * -> count (device int32), sel [count] = index into [B][nt], and pts4 / dirs / dists of fneus_outside_points for the listed samples
 * (buffers of B nt rows); alpha_full [B][nt] / rgb_full [B][nt][3] get zeros at the samples NOT listed (the listed ones are
 * written by fneus_outside_alpha_sel_fwd).  work: B + B nt int32.  fneus_nerf_bg_fwd / _bwd and fneus_dw_gemm_pp take `count`
 * as their device-side sample count. */
int fneus_outside_select(const float* rays_o, const float* rays_d, const float* z_core, const float* z_feed, int n_rays, int n,
                         int nt, float sample_dist, int32_t* work, float* pts4, float* dirs, float* dists, int32_t* sel,
                         int32_t* count, float* alpha_full, float* rgb_full, fneus_stream_t stream);
/* fneus_outside_alpha_fwd / _bwd over that list: row k < *count of density / rgb_raw / dists is sample sel[k] of the full arrays
 * (cap = rows of the list buffers) */
int fneus_outside_alpha_sel_fwd(const float* density, const float* rgb_raw, const float* dists, const int32_t* sel,
                                const int32_t* count, long cap, float* alpha_full, float* rgb_full, fneus_stream_t stream);
int fneus_outside_alpha_sel_bwd(const float* density, const float* rgb_full, const float* dists, const int32_t* sel,
                                const int32_t* count, long cap, const float* d_alpha_full /*or NULL*/,
                                const float* d_rgb_full /*or NULL*/, float* d_density /*[cap]*/, float* d_rgb_raw /*[cap][3]*/,
                                fneus_stream_t stream);
/* alpha = 1 - exp(-softplus(density) dist), rgb = sigmoid(rgb_raw) (renderer.py:137-138) and the adjoint (d_alpha / d_rgb may
 * be NULL = zero) */
int fneus_outside_alpha_fwd(const float* density, const float* rgb_raw, const float* dists, long n, float* alpha, float* rgb,
                            fneus_stream_t stream);
int fneus_outside_alpha_bwd(const float* density, const float* rgb, const float* dists, const float* d_alpha, const float* d_rgb,
                            long n, float* d_density, float* d_rgb_raw, fneus_stream_t stream);

/* ---- K5: NeuS SDF->alpha, front-to-back compositing, eikonal sums, first sign change
 *      (renderer.py:245-274, 290-293, 328-332, 360-372).  Per-ray outputs: color [B][3], wsum/wmax [B],
 *      eik [2][B] = (sum relax*(|g|-1)^2 ; sum relax), min_idx [B], sdf_mask [B] (u8), wpair [B][2] = inside-sphere
 *      weights at min_idx-1 / min_idx (0 when !sdf_mask).  inv_s is a device scalar: the value itself
 *      (inv_s_mode 0), or the `variance` parameter of SingleVarianceNetwork (inv_s_mode 1; the kernels then apply
 *      inv_s = clip(exp(10 variance), 1e-6, 1e6), fields.py:262-268 / renderer.py:245, and the backward returns the
 *      gradient with respect to `variance`).
 *      With bg_alpha / bg_color (womask, renderer.py:350-356) samples outside the unit sphere take the background
 *      NeRF's alpha / colour and n_out background samples are appended (weights then have n + n_out columns). */
int fneus_composite_fwd(const float* rays_o, const float* rays_d, const float* mid_z, const float* dists, const float* sdf,
                        const float* normal, const float* rgb, const float* inv_s, int inv_s_mode, int n_rays, int n,
                        float cos_anneal_ratio, const float* cos_anneal_dev /*device scalar overriding the float, or NULL*/,
                        const float* bg_alpha /*[B][n+n_out] or NULL*/,
                        const float* bg_color /*[B][n+n_out][3] or NULL*/, int n_out,
                        float* weights /*[B][n (+n_out)]*/, float* color, float* wsum, float* wmax, float* cdf,
                        float* inside, float* eik, int32_t* min_idx, unsigned char* sdf_mask, float* wpair,
                        const float* background_rgb /*[background_rows][3] (1 row or one per ray) or NULL: color +=
                        background_rgb (1 - wsum), renderer.py:367-368*/, int background_rows, fneus_stream_t stream);
/* adjoint of fneus_composite_fwd; d_weights may be NULL; d_inv_s is per ray (caller sums). */
int fneus_composite_bwd(const float* rays_o, const float* rays_d, const float* mid_z, const float* dists, const float* sdf,
                        const float* normal, const float* rgb, const float* inv_s, int inv_s_mode, int n_rays, int n,
                        float cos_anneal_ratio, const float* cos_anneal_dev /*device scalar overriding the float, or NULL*/,
                        const float* bg_alpha, const float* bg_color, int n_out,
                        const int32_t* min_idx, const unsigned char* sdf_mask,
                        const float* d_color, const float* d_wsum, const float* d_weights, const float* d_wpair,
                        const float* d_eiknum, float* d_sdf, float* d_normal, float* d_rgb, float* d_inv_s,
                        float* d_bg_alpha /*[B][n+n_out] or NULL*/, float* d_bg_color /*or NULL*/,
                        const float* background_rgb /*as in the forward: a constant*/, int background_rows, fneus_stream_t stream);

/* ---- optimiser: torch.optim.Adam.step() over the whole model in one launch (exp_runner.py:108, 179-181) ---------- */
/* segs: HOST array of contiguous parameter runs (32 per launch).  lr: device scalar.  step: device float[2] -- [0] the steps
 * taken so far (this update uses step[0] + 1 for the bias corrections and leaves it there), [1] scratch of the launch, zero
 * between calls (round 5: the increment was a launch of its own).  No weight decay, no amsgrad.  zero_grad != 0 clears each
 * gradient after use. */
int fneus_adam(const FneusAdamSegment* segs, int n_segs, const float* lr, float* step, double beta1, double beta2,
               double eps, int zero_grad, fneus_stream_t stream);

/* ---- box calibration (bench.py `box`; not part of the reference's path) ------------------------------------------------------------
 * fneus_probe_mfma: 1024 x 4 waves issue iters x 16 v_mfma_f32_32x32x16_bf16 back to back on the 32 KiB of bf16 operands given (random
 * data: the clock the chip holds depends on it); fneus_probe_mfma_flops(iters) FLOP per launch; ticks[0] = shader cycles, ticks[1] =
 * 100 MHz ticks of wave 0 over the loop.  fneus_probe_copy: float4 copy of `bytes` (a multiple of 16).                              */
int fneus_probe_mfma(const void* operands_32kib, int iters, unsigned long long* ticks /*[2]*/, float* sink /*[1]*/, fneus_stream_t stream);
long fneus_probe_mfma_flops(int iters);
int fneus_probe_copy(const void* src, void* dst, long bytes, fneus_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FNEUS_H */
