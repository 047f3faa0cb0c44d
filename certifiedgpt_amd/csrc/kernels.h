// kernels.h -- host-callable launchers of the HIP kernels (all enqueue on `stream`, no host sync).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cgpt {

typedef _Float16 half_t;

// ---------------------------------------------------------------------------------------------- GEMM
// C = A[M,K] * W[N,K]^T with a fused epilogue.  A/W are fp16, K contiguous.  Requirements (the library's
// own buffers satisfy them): K % 64 == 0; A readable for round_up(M,256) rows; W readable for
// round_up(N,256) rows; pad K-columns of A and W hold zeros.
enum GemmEpilogue {
    EPI_F16 = 0,        // out_f16[m,n] = acc + bias[n]
    EPI_F16_GELU = 1,   // out_f16[m,n] = gelu_erf(acc + bias[n])                        (eva_vit.py:60-61)
    EPI_F32 = 2,        // out_f32[m,n] = acc + bias[n]
    EPI_RESID = 3,      // out_f32[m,n] = aux_f32[m,n] + acc + bias[n]  (aux may alias out: x = x + f(x), eva_vit.py:180-181)
    EPI_PATCH = 4       // patch-embed: row m=(b,p) -> out_f32[b*(P+1)+1+p, n] = acc + bias[n] + aux[(1+p), n]  (eva_vit.py:209,337-340)
};
struct GemmParams {
    const half_t* A; int64_t lda;
    const half_t* W; int64_t ldw;
    const float* bias;            // [N] or nullptr
    void* out; int64_t ldo;
    const float* aux; int64_t ldaux;
    int M, N, K;
    int patches;                  // EPI_PATCH: P (patches per image)
    unsigned long long* dbg = nullptr;   // diagnostic builds only (-DCGPT_STAMPS): per-wave cycle sums
    unsigned long long* clk = nullptr;   // measurement hook (cgpt_profile_clock), 256-row kernels: += every workgroup's shader cycles [0] and
                                         // 100-MHz ticks [1] from its first to its last instruction; null = off
    int group_m = 8;              // tile-rows per group in the block->tile map (speed only)
    int ablate = 0;               // lab builds: 1 = no in-loop loads, 2 = no epilogue stores, 4 = no MFMAs; tests: see launch_gemm
};
hipError_t launch_gemm(int epilogue, const GemmParams& p, hipStream_t stream);
// gemm9.hip: 256x256 tile, two 32-MFMA phases per K-tile, operand parts requested 1.5 K-tiles ahead by LDS-DMA
hipError_t launch_v9_epi(int epilogue, const GemmParams& p, hipStream_t stream);
#ifdef CGPT_LAB
hipError_t launch_v6_epi(int epilogue, int mode, const GemmParams& p, hipStream_t stream);   // lab/gemm6.hip: 4-wave 128x128 wave tiles
hipError_t launch_v8_epi(int epilogue, const GemmParams& p, hipStream_t stream);             // lab/gemm8.hip: two 4-wave workgroups per CU, 128x256 tiles
#endif
extern int g_gemm_ablate;
extern int g_gemm_group_m;
extern int g_gemm_grid;      // 0 = one workgroup per CU; n > 0: at most n workgroups for the persistent 256x256 kernel (speed only)
extern unsigned long long* g_gemm_dbg;
extern int g_gemm_kernel;   // kernel override (speed only): 0 auto, 1 = 128x128, 3 = 256x128, 4 = 256x256 phased, 14 = 256x256 two-phase quadrant; lab builds: 2, 5..11, 15

// ---------------------------------------------------------------------------------------- attention
// O[b,q,h*hd + d] = sum_k softmax_k(scale * Q[b,q,h,:].K[b,k,h,:]) V[b,k,h,d]   (eva_vit.py:133-150,
// Qformer.py:244-264 with all-zero masks).  head_dim 88 or 64.  Row strides in elements.
struct AttnParams {
    const half_t* Q; int64_t ldq; int64_t q_batch_stride;
    const half_t* K; int64_t ldk;
    const half_t* V; int64_t ldv; int64_t kv_batch_stride;
    half_t* O; int64_t ldo; int64_t o_batch_stride;
    int B, heads, head_dim, Tq, Tk;
    float scale;
    unsigned long long* dbg = nullptr;   // diagnostic builds only (-DCGPT_STAMPS): per-wave phase cycle sums
};
hipError_t launch_attention(const AttnParams& p, hipStream_t stream);

// --------------------------------------------------------------------------------------- LayerNorm
// y = (x - mean) / sqrt(var + eps) * gamma + beta over D, fp32 statistics (base_model.py:281-287).
// Input row r is x + r*ldx; outputs y16 (fp16, may be null) and y32 (fp32, may be null).
// delta (fp16, may be null): pending residual update, applied first and written back: x += delta (eva_vit.py:180-181).
hipError_t launch_layernorm(float* x, int64_t ldx, const half_t* delta, int64_t ldd, const float* gamma,
                            const float* beta, float eps, half_t* y16, int64_t ldy16, float* y32, int64_t ldy32,
                            int64_t rows, int D, hipStream_t stream, const half_t* delta2 = nullptr, int64_t ldd2 = 0,
                            int keep_x = 0);   // delta2: second pending update; keep_x: do not write x + updates back
// x += delta without a LayerNorm (after the last block).  skip_mod > 0: rows r with r % skip_mod == 0 are left alone.
hipError_t launch_add_delta(float* x, int64_t ldx, const half_t* delta, int64_t ldd, int64_t rows, int D,
                            hipStream_t stream, int skip_mod = 0);

// ------------------------------------------------------------------------------ noise / im2col / misc
// smoothing.py:95-96 fused with the patch-embed im2col (eva_vit.py:202,209): for sample s = first_sample + b,
// patch p, element e=(c,i,j):  A[(b*P + p), e] = fp16(x[c, py*ps+i, px*ps+j] + sigma * eps_s[c, .., ..]).
// Batch row b carries sample first_sample + b for b < na and first_b + (b - na) otherwise (two index ranges per batch).
// per > 0: several clean images x[i] in one batch, image-major, `per` rows each (na of the first range, per - na of the
// second), both ranges shifted by i * img_stride sample indices for image i.
hipError_t launch_noise_im2col(const float* x, int img, int ps, int64_t first_sample, int na, int64_t first_b, int nb,
                               float sigma, uint64_t seed, half_t* A, int64_t lda, hipStream_t stream, int per = 0,
                               int64_t img_stride = 0, int64_t row0 = 0);   // row0: sequence row of batch row 0 (per > 0)
// Same im2col for caller-supplied images [nb,3,img,img] (no noise).
hipError_t launch_im2col(const float* images, int img, int ps, int nb, half_t* A, int64_t lda, hipStream_t stream);
// out[b, :] = x + sigma * eps_{first_sample + b}   (fp32 images; handle-free C-ABI cgpt_noise_batch)
hipError_t launch_noise_batch(const float* x, int64_t chw, int64_t first_sample, int64_t num, float sigma,
                              uint64_t seed, float* out, hipStream_t stream);
// RGF attack step (build-side rule, see elementwise.hip): out = clamp(x_adv + lr*sign(sum_i coeffs[i]*u_{first_dir+i}), x_clean +- eps);
// coeffs is a HOST array of q <= CGPT_RGF_MAX_DIRS floats.
#define CGPT_RGF_MAX_DIRS 32
hipError_t launch_rgf_step(const float* x_adv, const float* x_clean, int64_t chw, int64_t first_dir, int q, const float* coeffs,
                           float lr, float eps, uint64_t seed, float* out, hipStream_t stream);
// resid[b*T + 0, :] = cls + pos[0, :]   (eva_vit.py:337-340, CLS row)
hipError_t launch_cls_rows(const float* cls, const float* pos, float* resid, int64_t ld, int T, int nb, int D,
                           hipStream_t stream);
// dst32[b*rows + r, :] = src32[r, :], dst16 likewise (query_tokens.expand, minigpt4.py:132)
hipError_t launch_broadcast_rows(const float* src, int rows, int D, int nb, float* dst32, int64_t ld32,
                                 half_t* dst16, int64_t ld16, hipStream_t stream);
// dst16[b, :] = mean_r src32[b*rows + r, :]
hipError_t launch_mean_rows(const float* src, int64_t lds, int rows, int D, int nb, half_t* dst16, int64_t ld16,
                            hipStream_t stream);
// smoothing.py:97-98,101-105: counts[argmax_k logits[b,k]] += 1 (first maximal index), one wave per sample.
// Rows b < na vote into counts, rows b >= na into counts_b.
// per > 0 (several images, see launch_noise_im2col): image i = row / per votes into counts + i*2K / counts_b + i*2K.
hipError_t launch_vote(const float* logits, int64_t ld, int64_t num, int K, int64_t* counts, int64_t na, int64_t* counts_b,
                       hipStream_t stream, int per = 0, int64_t row0 = 0);
// Smooth.certify lines 46-56 (predict = 0) or Smooth.predict lines 73-79 (predict = 1) on device histograms; one wave.
// out[0] = label, out[1] = radius | p-value (float64).
hipError_t launch_finalize(const int64_t* csel, const int64_t* cest, int K, int64_t n, double alpha, double sigma, int predict,
                           double* out, hipStream_t stream);
// Fill a tensor from the counter-based normal stream: dst = mean + std * z (fp16 or fp32 destination, 2-D with ld).
hipError_t launch_fill_normal(void* dst, int is_f16, int64_t rows, int64_t cols, int64_t ld, float mean, float std,
                              uint64_t seed, uint64_t tensor_id, hipStream_t stream);
hipError_t launch_fill_const(float* dst, int64_t n, float v, hipStream_t stream);
// fp16 <-> fp32 2-D copies with leading dims (weight upload / download)
hipError_t launch_f32_to_f16(const float* src, int64_t lds, half_t* dst, int64_t ldd, int64_t rows, int64_t cols,
                             hipStream_t stream);
hipError_t launch_f16_to_f32(const half_t* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols,
                             hipStream_t stream);

// diag.hip: the dense fp16 MFMA rate this device sustains on random operands with nothing else running (synchronous diagnostic)
hipError_t run_mfma_sustained(double seconds, double* tflops, double* clock_ghz);

}  // namespace cgpt
