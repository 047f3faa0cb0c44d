/*
 * cgpt.h -- C-ABI of libcgpt.so: the MI355X-native randomized-smoothing certify/predict hot path.
 *
 * The reference (leodesouza/certifiedGPT) has no FFI: its hot path is the Python class
 * `Smooth` (randomized_smoothing/smoothing.py:13-117) calling a PyTorch base classifier
 * (MiniGPT4.encode_img, graphs/models/minigpt4/models/minigpt4.py:121-149, built from
 * eva_vit.py / Qformer.py).  Each entry point below names the reference interface it replaces.
 * A maintainer binds them with ctypes; certifiedgpt_amd/_lib.py is that binding.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures (streams are passed as void* = hipStream_t,
 *     0 = the null stream; PyTorch-ROCm's torch.cuda.current_stream().cuda_stream is accepted as is).
 *   - every function returns cgpt_status (0 = ok).  Nothing throws across the ABI;
 *     cgpt_last_error() returns a thread-local message for the last failing call.
 *   - the caller owns every buffer it passes.  `*_dev` pointers are device (HBM) pointers,
 *     `*_host` pointers are host pointers.
 *   - a handle is bound to one device and is NOT thread-safe: one handle per process per GPU.
 *     All device work is enqueued on the caller's stream; the only host syncs are inside
 *     cgpt_load_weight/cgpt_get_weight (host<->device copies) and where documented.
 *   - there is no CPU fallback: device entry points fail with CGPT_ERR_NO_DEVICE without a GPU.
 */
#ifndef CGPT_H
#define CGPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int cgpt_status;
enum {
    CGPT_OK = 0,
    CGPT_ERR_INVALID = 1,     /* bad argument / shape mismatch                         */
    CGPT_ERR_NO_DEVICE = 2,   /* no HIP device visible                                 */
    CGPT_ERR_HIP = 3,         /* a HIP runtime call failed (message in cgpt_last_error) */
    CGPT_ERR_NOT_FOUND = 4,   /* unknown weight name                                   */
    CGPT_ERR_STATE = 5        /* weights not loaded / wrong mode                       */
};

#define CGPT_ABSTAIN (-1)     /* Smooth.ABSTAIN, smoothing.py:17 */

/* What the base classifier computes after the noise is added (smoothing.py:97 `base_classifier(batch + noise)`). */
enum {
    CGPT_MODE_VIT_HEAD = 0,   /* EVA-ViT-G forward_features (eva_vit.py:332-349) -> ln_vision (base_model.py:281-287)
                                 on the CLS token -> build-side Linear(vit_dim -> num_classes) head.  BASELINE config 2. */
    CGPT_MODE_ENCODE_IMG = 1  /* full MiniGPT4.encode_img (minigpt4.py:121-149): ViT -> ln_vision -> Q-Former
                                 (Qformer.py:78-108,169-289,349-484) -> llama_proj, then a build-side head
                                 Linear(proj_dim -> num_classes) on the mean of the query tokens.             */
};

typedef struct cgpt_config {
    int32_t struct_size;      /* = sizeof(cgpt_config); guards ABI drift */
    int32_t mode;             /* CGPT_MODE_*                                                       */
    int32_t device;           /* HIP device ordinal                                                */
    int32_t num_classes;      /* Smooth.num_classes, smoothing.py:19                               */
    int32_t max_batch;        /* largest batch_size a call may use (sizes the HBM workspace)       */
    /* EVA-ViT (create_eva_vit_g, eva_vit.py:425-438) */
    int32_t img_size;         /* 224  */
    int32_t patch_size;       /* 14   */
    int32_t vit_dim;          /* 1408 */
    int32_t vit_depth;        /* 39   */
    int32_t vit_heads;        /* 16 (head_dim must be 88 or 64)                                    */
    int32_t vit_mlp;          /* 6144 = int(1408 * 4.3637)                                         */
    float   vit_ln_eps;       /* 1e-6 (eva_vit.py:436)                                             */
    float   ln_vision_eps;    /* 1e-5 (nn.LayerNorm default, base_model.py:281)                    */
    /* Q-Former (MiniGPT4.init_Qformer, minigpt4.py:90-119; BERT-base) -- ignored in CGPT_MODE_VIT_HEAD */
    int32_t qf_layers;        /* 12   */
    int32_t qf_dim;           /* 768  */
    int32_t qf_heads;         /* 12 (head_dim must be 64)                                          */
    int32_t qf_ffn;           /* 3072 */
    int32_t qf_queries;       /* 32   */
    int32_t qf_xattn_freq;    /* 2: cross-attention in layers 0,2,4,... (Qformer.py:386-395)       */
    float   qf_ln_eps;        /* 1e-12 */
    int32_t proj_dim;         /* 4096: llama_proj out features (minigpt4.py:76-78)                 */
} cgpt_config;

typedef struct cgpt_model* cgpt_handle;

/* ---- life cycle (replaces: MiniGPT4.__init__/from_config device placement, minigpt4.py:29-199) ---- */
cgpt_status cgpt_create(const cgpt_config* cfg, cgpt_handle* out);
cgpt_status cgpt_destroy(cgpt_handle h);
const char* cgpt_last_error(void);
const char* cgpt_version(void);

/* ---- weights (replaces: load_state_dict + convert_weights_to_fp16, eva_vit.py:407-422,454) ----
 * Names follow the reference state_dict:  "visual_encoder.patch_embed.proj.weight", "visual_encoder.cls_token",
 * "visual_encoder.pos_embed", "visual_encoder.blocks.<i>.{norm1,norm2}.{weight,bias}",
 * "visual_encoder.blocks.<i>.attn.{qkv.weight,q_bias,v_bias,proj.weight,proj.bias}",
 * "visual_encoder.blocks.<i>.mlp.{fc1,fc2}.{weight,bias}", "ln_vision.{weight,bias}", "query_tokens",
 * "Qformer.bert.embeddings.LayerNorm.{weight,bias}",
 * "Qformer.bert.encoder.layer.<i>.{attention,crossattention}.self.{query,key,value}.{weight,bias}",
 * "Qformer.bert.encoder.layer.<i>.{attention,crossattention}.output.{dense,LayerNorm}.{weight,bias}",
 * "Qformer.bert.encoder.layer.<i>.{intermediate_query,output_query}.dense.{weight,bias}",
 * "Qformer.bert.encoder.layer.<i>.output_query.LayerNorm.{weight,bias}", "llama_proj.{weight,bias}",
 * and the build-side "head.{weight,bias}".
 * `data_host` is float32, contiguous, in the PyTorch shape; `numel` must match.  GEMM weights are stored
 * in HBM as fp16 (the reference's precision contract on HIP: fp16 weights + fp16 autocast,
 * eva_vit.py:407-414, base_model.py:132-142); LayerNorm / bias / pos_embed parameters stay fp32. */
cgpt_status cgpt_load_weight(cgpt_handle h, const char* name, const float* data_host, int64_t numel);
/* Copy a weight back as float32 (exactly the values the device computes with, i.e. after fp16 rounding). */
cgpt_status cgpt_get_weight(cgpt_handle h, const char* name, float* out_host, int64_t numel);
/* Number of elements of a named weight (for sizing cgpt_get_weight buffers); <0 if unknown. */
int64_t     cgpt_weight_numel(cgpt_handle h, const char* name);
/* Name of the i-th weight (NULL past the end) -- lets a binding enumerate the state_dict. */
const char* cgpt_weight_name(cgpt_handle h, int32_t index);
/* Fill every weight on the device following the reference's init law (eva_vit.py:295-323:
 * trunc_normal(.02) Linear weights, zero biases, LN (1,0), proj/fc2 / sqrt(2*layer_id); Qformer.py:664-674:
 * normal(0,.02); minigpt4.py:99-102) from a counter-based generator.  There is no checkpoint in the container. */
cgpt_status cgpt_init_synthetic_weights(cgpt_handle h, uint64_t seed, void* stream);

/* ---- the hot loop: Smooth._sample_noise (smoothing.py:81-99) ----
 * For global sample indices s in [first_sample, first_sample + num): draw eps_s ~ N(0,1)^{3xHxW} from the
 * counter-based stream keyed (noise_seed, s, element), run the base classifier on x + sigma*eps_s in
 * batches of at most batch_size, take argmax(1) and ADD the votes into counts_dev[num_classes] (int64).
 * counts_dev is NOT zeroed here (callers accumulate shards); no host sync.
 * Sharding-aware through first_sample: counts are bit-identical for any partition of the index range. */
cgpt_status cgpt_sample_counts(cgpt_handle h, const float* x_dev, int64_t first_sample, int64_t num,
                               int64_t batch_size, float sigma, uint64_t noise_seed,
                               int64_t* counts_dev, void* stream);
/* Two sample-index ranges in ONE pass: Smooth.certify draws n0 selection samples and n estimation samples
 * (smoothing.py:44,48); they are independent, so both can ride in the same classifier batches.  Samples
 * [first_a, first_a+num_a) vote into counts_a_dev, samples [first_b, first_b+num_b) into counts_b_dev; batches of up to
 * batch_size samples may span the two ranges.  Results are identical to two cgpt_sample_counts calls. */
cgpt_status cgpt_sample_counts2(cgpt_handle h, const float* x_dev, int64_t first_a, int64_t num_a, int64_t* counts_a_dev,
                                int64_t first_b, int64_t num_b, int64_t* counts_b_dev, int64_t batch_size, float sigma,
                                uint64_t noise_seed, void* stream);
/* The same pass for SEVERAL images at once: image i (x_dev + i*3*H*W) draws samples first_a + i*image_stride + [0, num_a)
 * and first_b + i*image_stride + [0, num_b), exactly what num_images consecutive Smooth.certify calls would use with
 * image_stride = n0 + n.  counts_dev is int64 [num_images, 2, num_classes] (selection row, estimation row), ADDED into.
 * The (image, sample) rows of all images form one sequence that is cut into classifier batches of max_batch rows, NOT aligned
 * to image boundaries: a rank that owns only a thin slice of every image's samples (N/8 on 8 GPUs) still runs full batches,
 * and max_batch can be chosen for the GEMMs' tile quantisation (255 samples = 65 535 token rows = exactly 256 tile rows)
 * rather than for n0 + n.  Counts are bit-identical to the per-image calls. */
cgpt_status cgpt_sample_counts_images(cgpt_handle h, const float* x_dev, int64_t num_images, int64_t first_a, int64_t num_a,
                                      int64_t first_b, int64_t num_b, int64_t image_stride, int64_t* counts_dev, float sigma,
                                      uint64_t noise_seed, void* stream);
/* Same batches, but return the logits [num, num_classes] float32 instead of voting (parity tests; num <= max_batch). */
cgpt_status cgpt_forward_logits(cgpt_handle h, const float* x_dev, int64_t first_sample, int64_t num,
                                float sigma, uint64_t noise_seed, float* logits_dev, void* stream);
/* Run the classifier on caller-supplied images [num,3,H,W] float32 (no noise) -> logits [num,num_classes]. */
cgpt_status cgpt_classify(cgpt_handle h, const float* images_dev, int64_t num, float* logits_dev, void* stream);
/* MiniGPT4.encode_img (minigpt4.py:121-149) as a first-class product: ViT -> ln_vision -> Q-Former -> llama_proj for
 * `num` caller-supplied images [num,3,H,W] float32, written to inputs_llama_dev [num, qf_queries, proj_dim] float32 -- the
 * tensor MiniGPTBase.generate splices into the LLM prompt (minigpt_base.py:401-405; `atts_llama` is all ones, minigpt4.py:148,
 * and is not materialised).  Any num >= 0 (internally cut into batches of max_batch); CGPT_MODE_ENCODE_IMG handles only
 * (CGPT_ERR_STATE otherwise); no host sync.  The build-side label head is not evaluated. */
cgpt_status cgpt_encode_img(cgpt_handle h, const float* images_dev, int64_t num, float* inputs_llama_dev, void* stream);
/* The same for the Monte-Carlo draws of Smooth._sample_noise (smoothing.py:95-97 with a generating base classifier):
 * row b of inputs_llama_dev is encode_img(x + sigma * eps_{first_sample + b}); the noise is generated inside the patch-embed
 * operand as in cgpt_sample_counts, so the noisy images never exist in HBM. */
cgpt_status cgpt_encode_img_noisy(cgpt_handle h, const float* x_dev, int64_t first_sample, int64_t num, float sigma,
                                  uint64_t noise_seed, float* inputs_llama_dev, void* stream);
/* Intermediate activations of the last cgpt_forward_logits / cgpt_classify call, as float32, for parity tests:
 * what = "vit_out"   [num, T, vit_dim]   (VisionTransformer.forward_features output, eva_vit.py:349)
 *        "ln_vision" [num, T, vit_dim]   (CGPT_MODE_ENCODE_IMG only; minigpt4.py:129)
 *        "qformer"   [num, Q, qf_dim]    (query_output.last_hidden_state, minigpt4.py:134-139)
 *        "llama"     [num, Q, proj_dim]  (inputs_llama, minigpt4.py:141)                                      */
cgpt_status cgpt_get_activation(cgpt_handle h, const char* what, float* out_dev, int64_t numel, void* stream);

/* ---- pieces of the loop, for callers whose base classifier is not this library's
 *      (e.g. full MiniGPT-4 with a Vicuna decode on PyTorch-ROCm, minigpt_base.py:374-448) ---- */
/* smoothing.py:95-96: out[b] = x + sigma * eps_{first_sample+b}; out_dev is [num,3,H,W] float32.  Handle-free. */
cgpt_status cgpt_noise_batch(const float* x_dev, int64_t chw, int64_t first_sample, int64_t num, float sigma,
                             uint64_t noise_seed, float* out_dev, void* stream);
/* One step of the random-gradient-free black-box attack of BASELINE configs[4] (the reference has no code for it, only prose:
 * README.md:62-64,108-120; the rule is this library's): with u_s the N(0,1) image of sample index s of the noise stream (the
 * direction cgpt_noise_batch(x, s, 1, delta, seed) adds),
 *   out = clamp(x_adv + lr * sign(sum_{i<num_dirs} coeffs_host[i] * u_{first_dir+i}), x_clean - eps, x_clean + eps).
 * coeffs_host: num_dirs (1..32) finite-difference coefficients on the HOST; the three images are [chw] float32 on the device
 * (out_dev may alias x_adv_dev).  Handle-free. */
cgpt_status cgpt_rgf_step(const float* x_adv_dev, const float* x_clean_dev, int64_t chw, int64_t first_dir, int32_t num_dirs,
                          const float* coeffs_host, float lr, float eps, uint64_t noise_seed, float* out_dev, void* stream);
/* smoothing.py:97-98,101-105: counts_dev[argmax(logits[b,:])] += 1 for b < num (first max index on ties). */
cgpt_status cgpt_vote(const float* logits_dev, int64_t num, int32_t num_classes, int64_t* counts_dev, void* stream);

/* ---- the one collective of the path, for callers that are not Python (Python callers use torch.distributed, whose "nccl"
 *      backend IS RCCL on ROCm): in-place sum over all ranks of the int64 vote histograms a cgpt_sample_counts* call produced
 *      on each rank's shard of the sample range -- ncclAllReduce(counts, counts, count, ncclInt64, ncclSum, comm, stream) over
 *      xGMI.  rccl_comm is the caller's ncclComm_t (one process per GPU); count = num_classes for one histogram, or
 *      num_images * 2 * num_classes for the table of cgpt_sample_counts_images.  Enqueued on `stream`, no host sync.
 *      (The reference has no call site: its collectives are torch_xla's)
 *      SAME-INSTANCE REQUIREMENT: a communicator lives inside the RCCL library instance that created it.  libcgpt.so does not
 *      link RCCL and never loads one: cgpt_allreduce_counts binds to the ONE librccl already mapped in the process (whatever its
 *      dlopen flags) and returns CGPT_ERR_STATE when none or more than one is mapped (e.g. torch's bundled copy next to
 *      /opt/rocm's); in that case, or whenever you hold the pointer anyway, pass the ncclAllReduce of the library that created
 *      rccl_comm to cgpt_allreduce_counts_fn.  Stateless and callable from any thread. ---- */
cgpt_status cgpt_allreduce_counts(void* rccl_comm, int64_t* counts_dev, int64_t count, void* stream);
cgpt_status cgpt_allreduce_counts_fn(void* nccl_allreduce, void* rccl_comm, int64_t* counts_dev, int64_t count, void* stream);

/* ---- statistics: pure host functions, float64, no device needed ----
 * Smooth.certify lines 46-56 given the two histograms (smoothing.py:44,48). */
cgpt_status cgpt_certify_from_counts(const int64_t* counts_selection, const int64_t* counts_estimation,
                                     int32_t num_classes, int64_t n, double alpha, double sigma,
                                     int32_t* label_out, double* radius_out);
/* The same for a table of histograms counts[num_images][2][num_classes] (selection then estimation per image: what
 * cgpt_sample_counts_images fills), one call for the group of images of Smooth.certify_many: image i gets exactly what
 * cgpt_certify_from_counts gives for its two rows (the per-image Python round trips were 70 us per image of host time in which the
 * GPU idles: 3 % of a rank's time at 8 GPUs). */
cgpt_status cgpt_certify_many_from_counts(const int64_t* counts, int64_t num_images, int32_t num_classes, int64_t n, double alpha,
                                          double sigma, int32_t* labels_out, double* radii_out);
/* Smooth.predict lines 73-79 given the histogram (smoothing.py:72). */
cgpt_status cgpt_predict_from_counts(const int64_t* counts, int32_t num_classes, double alpha, int32_t* label_out);
/* The same two decisions computed ON THE DEVICE from device histograms (no histogram copy, no host sync): one wavefront,
 * arg-max / top-2 by xor-shuffles, Clopper-Pearson bound, binomial test and Phi^-1 in float64 with the same code as the
 * host functions.  out2_dev[0] = label (-1 = abstain) as a double, out2_dev[1] = radius (certify) or p-value (predict). */
cgpt_status cgpt_certify_device(const int64_t* counts_selection_dev, const int64_t* counts_estimation_dev, int32_t num_classes,
                                int64_t n, double alpha, double sigma, double* out2_dev, void* stream);
cgpt_status cgpt_predict_device(const int64_t* counts_dev, int32_t num_classes, double alpha, double* out2_dev, void* stream);
/* Smooth._lower_confidence_bound (smoothing.py:107-117) == statsmodels proportion_confint(NA,N,2*alpha,"beta")[0]. */
double cgpt_lower_confidence_bound(int64_t NA, int64_t N, double alpha);
/* scipy.stats.binom_test(x, n, p) two-sided (scipy 1.7 algorithm; call site smoothing.py:76). */
double cgpt_binom_test(int64_t x, int64_t n, double p);
/* scipy.stats.norm.ppf (call site smoothing.py:55). */
double cgpt_norm_ppf(double p);

/* ---- measurement hooks (bench.py roofline) ----
 * When enabled, every GEMM launch is bracketed by HIP events on the launch stream; totals are read back with
 * cgpt_profile_read (which synchronises those events).  kind: 0 = all GEMMs (reading it drains the log), 1 = the ViT MLP fc1
 * (GELU epilogue) GEMMs only, 2 = ViT qkv, 3 = ViT attention proj, 4 = ViT MLP fc2. */
cgpt_status cgpt_profile_enable(cgpt_handle h, int32_t on);
cgpt_status cgpt_profile_read(cgpt_handle h, int32_t kind, double* total_ms, double* total_flops, int64_t* launches);
/* The shader clock the profiled GEMMs of `kind` (as above; 0 = all) actually ran at: every workgroup of the 256-row kernels reads
 * s_memtime and s_memrealtime at its first and last instruction while profiling is on; clock_ghz = sum of cycles / sum of 100-MHz
 * ticks x 0.1 (0 when nothing was profiled).  Synchronises the device.  Read it BEFORE cgpt_profile_read(kind 0), which resets
 * the sums with the log.  For ONE run, frac = (share of cycles in which the matrix pipes issue) x this clock / 2.4 GHz, so a clock
 * give-back shows up here; bench.py prints that share as roofline.implied_mfma_busy_frac.  (roofline.pmc.mfma_busy_frac and .clock_ghz
 * come from a separate rocprofv3 --pmc run, which holds other clocks: never multiply numbers of the two runs.) */
cgpt_status cgpt_profile_clock(cgpt_handle h, int32_t kind, double* clock_ghz);
/* The classifier batches the handle ran since profiling was last switched on, in order: samples (rows of smoothing.py:93-97's `batch`)
 * of every base-classifier forward, i.e. how cgpt_sample_counts* cut their sample ranges (smoothing.py:91-98: `this_batch_size`).
 * *count_out = number of forwards; the first min(count, capacity) sizes are written to samples_out (capacity 0: count only).  Host
 * bookkeeping only, nothing is synchronised.  bench.py prints it per rank, so that a multi-GPU line shows the GEMM shapes it ran. */
cgpt_status cgpt_profile_batches(cgpt_handle h, int32_t* samples_out, int64_t capacity, int64_t* count_out);

/* Process-wide SPEED knobs.  No option changes a result: every accepted value gives bit-identical outputs (tested).
 * THREADING OF THE LIBRARY (not only of a handle): these options AND the per-device first-launch caches of the kernel launchers (LDS
 * attribute + CU count, set on a device's first launch of each kernel) are process-global and unsynchronised.  One thread drives the
 * library until every device in use has run its first forward; afterwards handles on DIFFERENT devices may be driven from different
 * threads (one thread per handle), and options are set only while no launch is in flight.  The supported deployment is the reference's:
 * one process per GPU (launch.py:110-120).
 *   "gemm_kernel": 0 = automatic choice (default); 1 = 128x128 register-staged tile, 3 = 256x128 direct-to-LDS tile,
 *                  4 = 256x256 phase-alternating tile, 14 = 256x256 two-phase quadrant tile with a 1.5-K-tile LDS-DMA run-ahead (the
 *                  automatic choice for M >= 1024 when the shape has more than 128 tiles of 256x256).
 *   "gemm_ablate": TEST-ONLY bit mask; each bit turns ONE optimisation of the 256x256 kernels off without changing a result, so that
 *                  the test suite can check that the bits do not depend on it: 512 = LDS-transposed fp16 epilogue, 16384 = 192-column
 *                  last tiles for N = 256k+128.  Any other bit is rejected with CGPT_ERR_INVALID.  Process-global and unsynchronised: set it only while no launch is in flight.
 *   "gemm_grid":   0 (default) = one workgroup of the persistent 256x256 GEMM per CU; n > 0 (a multiple of 8) = at most n workgroups, i.e.
 *                  CUs left free for kernels of another stream (tools/two_stream_probe.py; measured: no split beats the default).
 *   "sync_batches": MEASUREMENT aid; 1 = cgpt_sample_counts* wait for the stream after every classifier batch (a --pmc profiler keeps a
 *                  record per in-flight dispatch and one call can enqueue tens of thousands: profiles/r06/pmc_sigsegv.txt); 0 (default) =
 *                  nothing is synchronised.  It stays in the shipped library on purpose: counters must be taken on the binary that ships.
 *   "trace_batches": MEASUREMENT aid; 1 = one stderr line per classifier batch the cgpt_sample_counts* entry points enqueue; 0 (default).
 * (A lab build of the library -- make LAB=1, never shipped -- additionally accepts the experimental schedules 2, 5..11, 15 and
 * timing-study switches that skip work; profiles/r01/gemm_variants.txt.) */
cgpt_status cgpt_set_option(const char* key, int32_t value);

/* Measurement aid (no counterpart in the reference): the dense fp16 MFMA rate THIS device sustains on random operands when it does nothing
 * else -- v_mfma_f32_16x16x32_f16 back to back from registers for `seconds` (clock settled first), in TFLOP/s, and the in-kernel shader
 * clock in GHz.  The data sheet's 2.5 PFLOP/s assumes 2.4 GHz; under an MFMA-dense load an MI355X holds 1.8-1.95 GHz, so bench.py reports
 * a kernel's fraction of the data-sheet peak AND of this rate (profiles/r04/mfma_sustained.txt).  Synchronous; runs on the null stream of
 * the CURRENT device: select the device first (hipSetDevice / torch.cuda.set_device).  Every HIP call inside is checked: a failed launch,
 * a zero elapsed time or an unwritten clock buffer returns an error (outputs 0), never a number. */
cgpt_status cgpt_mfma_sustained(double seconds, double* tflops_out, double* clock_ghz_out);

/* ---- raw kernels exported for unit tests and reuse (all fp16 operands are IEEE binary16) ----
 * C[M,N] = A[M,K] * W[N,K]^T (+ bias[N]) ; A row stride lda, W row stride ldw (elements), fp32 accumulate.
 * Requirements: A must be readable for ceil(M/256)*256 rows and W for ceil(N/256)*256 rows (zero padded; the
 * library's own buffers always are); K%64==0.  out is fp32 [M,ldc]. */
cgpt_status cgpt_gemm_f16(const void* A_dev, int64_t lda, const void* W_dev, int64_t ldw, const float* bias_dev,
                          float* C_dev, int64_t ldc, int64_t M, int64_t N, int64_t K, void* stream);
/* nn.Linear with a fused epilogue: 0 = fp16 out (acc + bias), 1 = fp16 out gelu_erf(acc + bias), 2 = fp32 out,
 * 3 = fp32 out = aux + acc + bias (aux may alias out: the residual add x = x + f(x), eva_vit.py:180-181). */
cgpt_status cgpt_linear_f16(const void* A_dev, int64_t lda, const void* W_dev, int64_t ldw, const float* bias_dev,
                            void* out_dev, int64_t ldo, const float* aux_dev, int64_t ldaux, int64_t M, int64_t N, int64_t K,
                            int32_t epilogue, void* stream);
/* softmax(scale * Q K^T) V per (batch, head): Q [B,Tq,ldq] K,V [B,Tk,ldkv] fp16 with head h at column h*head_dim;
 * O [B,Tq,ldo] fp16.  head_dim in {64, 88}.  (eva_vit.py:133-150; Qformer.py:244-264 with zero masks)
 * Tk <= 288: K/V of a head resident in LDS; Tk > 288 (448^2 images): K/V streamed in 288-key chunks, online softmax. */
cgpt_status cgpt_attention_f16(const void* Q_dev, int64_t ldq, const void* K_dev, const void* V_dev, int64_t ldkv,
                               void* O_dev, int64_t ldo, int32_t B, int32_t heads, int32_t head_dim,
                               int32_t Tq, int32_t Tk, float scale, void* stream);
/* y = LayerNorm(x) * gamma + beta over the last dim D (fp32 statistics), x fp32 [rows, ldx] -> y fp16 [rows, ldy]
 * (and optionally y32 fp32 [rows, ldy32] when y32_dev != NULL).  (eva_vit.py:162,168; base_model.py:281-287) */
cgpt_status cgpt_layernorm(const float* x_dev, int64_t ldx, const float* gamma_dev, const float* beta_dev, float eps,
                           void* y_dev, int64_t ldy, float* y32_dev, int64_t ldy32, int64_t rows, int32_t D,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CGPT_H */
