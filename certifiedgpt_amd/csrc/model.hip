// model.hip -- the handle behind include/cgpt.h: weights in HBM, the workspace, and the launch sequence of one
// base-classifier forward (what `self.base_classifier(batch + noise)` does at smoothing.py:97 when the classifier
// is MiniGPT-4's image encoder, minigpt4.py:121-149).
//
// HBM layout (one handle = one GPU; everything stays resident, nothing is re-allocated per call):
//   weights    fp16 matrices [round_up(N,256)][round_up(K,64)] zero padded (nn.Linear layout, K contiguous);
//              fp32 vectors for biases / LayerNorm affine / cls_token / pos_embed / query_tokens.
//   workspace  sized for max_batch samples: im2col A [nb*P][640] fp16, residual stream [nb*T][D] fp32,
//              LayerNorm output / attention output [nb*T][D] fp16, qkv [nb*T][3D] fp16, MLP hidden [nb*T][6144] fp16, ...
//              Row counts are padded to 256 and K-dims to 64 so every GEMM tile load is in bounds; pad columns
//              are zero at allocation and never written.
#include <dlfcn.h>
#include <link.h>
#include <string.h>
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/cgpt.h"
#include "common_host.h"
#include "kernels.h"

using namespace cgpt;

// ------------------------------------------------------------------------------------------- errors
static thread_local std::string g_last_error;
cgpt_status cgpt_fail(cgpt_status code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return cgpt_fail(CGPT_ERR_HIP, std::string(#expr) + " -> " + hipGetErrorString(e_));       \
    } while (0)
#define CGCHK(expr)                         \
    do {                                    \
        cgpt_status s_ = (expr);            \
        if (s_ != CGPT_OK) return s_;       \
    } while (0)

static inline int64_t ru(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------- model
namespace {

enum InitKind { INIT_ZERO = 0, INIT_ONE = 1, INIT_NORMAL = 2 };

struct View {          // a named parameter: a [rows, cols] window of a device buffer
    std::string name;
    void* base = nullptr;
    bool f16 = false;
    int64_t rows = 0, cols = 0, ld = 0, off = 0;
    int init = INIT_ZERO;
    float init_std = 0.f;
};

struct VitLayer {
    float *n1w, *n1b, *n2w, *n2b, *bqkv, *bproj, *bfc1, *bfc2;
    half_t *Wqkv, *Wproj, *Wfc1, *Wfc2;
};
struct QfLayer {
    half_t *Wqkv, *Wo, *Wxq, *Wxo, *Wi, *Wo2;
    float *bqkv, *bo, *lnw, *lnb, *bxq, *bxo, *xlnw, *xlnb, *bi, *bo2, *ln2w, *ln2b;
    int xattn_index;   // -1 when the layer has no cross-attention
};
struct ProfEvent { hipEvent_t a, b; double flops; int kind; };

}  // namespace

struct cgpt_model {
    cgpt_config cfg;
    int P, T, D, Dk, Kpatch, Kpatch_p, mlp, mlp_k, K;           // ViT derived dims
    int H, Hk, F, Fk, Q, nx, PD, PDk;                            // Q-Former derived dims
    int64_t ld_qkv, ld_kv;
    std::vector<void*> allocs;
    std::vector<View> views;
    std::map<std::string, int> index;
    // weights
    float *cls, *pos, *bpatch, *lnvw, *lnvb, *qtok, *qembw, *qembb, *bkv_all, *bproj_l, *bhead;
    half_t *Wpatch, *Wkv_all, *Wproj_l, *Whead;
    std::vector<VitLayer> vit;
    std::vector<QfLayer> qf;
    // workspace
    half_t *Apatch, *xn, *qkv, *attn, *hid, *delta, *delta2, *cls16, *emb, *kv_all, *qh16, *qqkv, *qctx, *qq, *qff, *pooled;
    float *resid, *logits, *qemb0, *qh32, *qtmp, *llama;
    int last_nb = 0;
    bool pending_delta = false;   // CGPT_MODE_VIT_HEAD: the last block's fc2 output has been added to the CLS rows only
    bool profile = false;
    std::vector<ProfEvent> events;
    std::vector<hipEvent_t> event_pool;      // timing-only events (hipEventDisableSystemFence), reused across profile_read(0) drains
    unsigned long long* clk_dev = nullptr;   // [8][2]: per GEMM kind, shader cycles and 100-MHz ticks summed over workgroups (cgpt_profile_clock)
    std::vector<int32_t> batch_log;          // samples of every classifier forward while profiling is on (cgpt_profile_batches)
};

namespace {

cgpt_status dev_alloc(cgpt_model* m, size_t bytes, void** out) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, bytes ? bytes : 16));
    HIPCHK(hipMemset(p, 0, bytes ? bytes : 16));
    m->allocs.push_back(p);
    *out = p;
    return CGPT_OK;
}

void add_view(cgpt_model* m, const std::string& name, void* base, bool f16, int64_t rows, int64_t cols, int64_t ld,
              int64_t off, int init, float std_) {
    View v;
    v.name = name; v.base = base; v.f16 = f16; v.rows = rows; v.cols = cols; v.ld = ld; v.off = off;
    v.init = init; v.init_std = std_;
    m->index[name] = (int)m->views.size();
    m->views.push_back(v);
}

// fp16 GEMM weight [N,K] (zero padded to [ru(N,256)][ru(K,64)])
cgpt_status new_mat(cgpt_model* m, int64_t N, int64_t Kd, half_t** out) {
    void* p;
    CGCHK(dev_alloc(m, (size_t)ru(N, 256) * ru(Kd, 64) * sizeof(half_t), &p));
    *out = (half_t*)p;
    return CGPT_OK;
}
cgpt_status new_vec(cgpt_model* m, int64_t n, float** out) {
    void* p;
    CGCHK(dev_alloc(m, (size_t)ru(n, 128) * sizeof(float), &p));
    *out = (float*)p;
    return CGPT_OK;
}
cgpt_status add_mat(cgpt_model* m, const std::string& name, int64_t N, int64_t Kd, float std_, half_t** out) {
    CGCHK(new_mat(m, N, Kd, out));
    add_view(m, name, *out, true, N, Kd, ru(Kd, 64), 0, INIT_NORMAL, std_);
    return CGPT_OK;
}
cgpt_status add_vec(cgpt_model* m, const std::string& name, int64_t n, int init, float std_, float** out) {
    CGCHK(new_vec(m, n, out));
    add_view(m, name, *out, false, 1, n, n, 0, init, std_);
    return CGPT_OK;
}
// activation buffers
cgpt_status new_act16(cgpt_model* m, int64_t rows, int64_t cols, half_t** out) {
    void* p;
    CGCHK(dev_alloc(m, (size_t)ru(rows, 256) * ru(cols, 64) * sizeof(half_t), &p));
    *out = (half_t*)p;
    return CGPT_OK;
}
cgpt_status new_act32(cgpt_model* m, int64_t rows, int64_t cols, float** out) {
    void* p;
    CGCHK(dev_alloc(m, (size_t)ru(rows, 256) * cols * sizeof(float), &p));
    *out = (float*)p;
    return CGPT_OK;
}

cgpt_status build(cgpt_model* m) {
    const cgpt_config& c = m->cfg;
    m->P = (c.img_size / c.patch_size) * (c.img_size / c.patch_size);
    m->T = m->P + 1;
    m->D = c.vit_dim; m->Dk = (int)ru(m->D, 64);
    m->Kpatch = 3 * c.patch_size * c.patch_size; m->Kpatch_p = (int)ru(m->Kpatch, 64);
    m->mlp = c.vit_mlp; m->mlp_k = (int)ru(m->mlp, 64);
    m->K = c.num_classes;
    const bool full = c.mode == CGPT_MODE_ENCODE_IMG;
    m->H = c.qf_dim; m->Hk = (int)ru(m->H, 64); m->F = c.qf_ffn; m->Fk = (int)ru(m->F, 64); m->Q = c.qf_queries;
    m->PD = c.proj_dim; m->PDk = (int)ru(m->PD, 64);
    m->nx = 0;
    if (full) for (int i = 0; i < c.qf_layers; ++i) if (i % c.qf_xattn_freq == 0) m->nx++;
    const int D = m->D, T = m->T, H = m->H;

    // ---- weights, registered in the order of oracle/model_oracle.py:param_shapes (tensor id = index)
    const float S = 0.02f;
    CGCHK(add_vec(m, "visual_encoder.cls_token", D, INIT_NORMAL, S, &m->cls));
    {
        void* p; CGCHK(dev_alloc(m, (size_t)T * D * sizeof(float), &p)); m->pos = (float*)p;
        add_view(m, "visual_encoder.pos_embed", p, false, T, D, D, 0, INIT_NORMAL, S);
    }
    CGCHK(add_mat(m, "visual_encoder.patch_embed.proj.weight", D, m->Kpatch, S, &m->Wpatch));
    CGCHK(add_vec(m, "visual_encoder.patch_embed.proj.bias", D, INIT_ZERO, 0, &m->bpatch));
    m->vit.resize(c.vit_depth);
    for (int i = 0; i < c.vit_depth; ++i) {
        VitLayer& L = m->vit[i];
        const std::string b = "visual_encoder.blocks." + std::to_string(i) + ".";
        const float div = 1.0f / sqrtf(2.0f * (float)(i + 1));            // eva_vit.py:308-314
        CGCHK(add_vec(m, b + "norm1.weight", D, INIT_ONE, 0, &L.n1w));
        CGCHK(add_vec(m, b + "norm1.bias", D, INIT_ZERO, 0, &L.n1b));
        CGCHK(new_vec(m, 3 * D, &L.bqkv));                                // cat(q_bias, 0, v_bias), eva_vit.py:127
        add_view(m, b + "attn.q_bias", L.bqkv, false, 1, D, D, 0, INIT_ZERO, 0);
        add_view(m, b + "attn.v_bias", L.bqkv, false, 1, D, D, 2 * D, INIT_ZERO, 0);
        CGCHK(add_mat(m, b + "attn.qkv.weight", 3 * D, D, S, &L.Wqkv));
        CGCHK(add_mat(m, b + "attn.proj.weight", D, D, S * div, &L.Wproj));
        CGCHK(add_vec(m, b + "attn.proj.bias", D, INIT_ZERO, 0, &L.bproj));
        CGCHK(add_vec(m, b + "norm2.weight", D, INIT_ONE, 0, &L.n2w));
        CGCHK(add_vec(m, b + "norm2.bias", D, INIT_ZERO, 0, &L.n2b));
        CGCHK(add_mat(m, b + "mlp.fc1.weight", m->mlp, D, S, &L.Wfc1));
        CGCHK(add_vec(m, b + "mlp.fc1.bias", m->mlp, INIT_ZERO, 0, &L.bfc1));
        CGCHK(add_mat(m, b + "mlp.fc2.weight", D, m->mlp, S * div, &L.Wfc2));
        CGCHK(add_vec(m, b + "mlp.fc2.bias", D, INIT_ZERO, 0, &L.bfc2));
    }
    CGCHK(add_vec(m, "ln_vision.weight", D, INIT_ONE, 0, &m->lnvw));
    CGCHK(add_vec(m, "ln_vision.bias", D, INIT_ZERO, 0, &m->lnvb));
    if (full) {
        {
            void* p; CGCHK(dev_alloc(m, (size_t)m->Q * H * sizeof(float), &p)); m->qtok = (float*)p;
            add_view(m, "query_tokens", p, false, m->Q, H, H, 0, INIT_NORMAL, S);
        }
        CGCHK(add_vec(m, "Qformer.bert.embeddings.LayerNorm.weight", H, INIT_ONE, 0, &m->qembw));
        CGCHK(add_vec(m, "Qformer.bert.embeddings.LayerNorm.bias", H, INIT_ZERO, 0, &m->qembb));
        // K/V projections of all cross-attention layers share their input (image_embeds): one [nx*2H, D] weight
        CGCHK(new_mat(m, (int64_t)m->nx * 2 * H, D, &m->Wkv_all));
        CGCHK(new_vec(m, (int64_t)m->nx * 2 * H, &m->bkv_all));
        m->qf.resize(c.qf_layers);
        int xi = 0;
        for (int i = 0; i < c.qf_layers; ++i) {
            QfLayer& L = m->qf[i];
            const std::string lp = "Qformer.bert.encoder.layer." + std::to_string(i) + ".";
            // self-attention: query/key/value fused into one [3H, H] GEMM
            CGCHK(new_mat(m, 3 * H, H, &L.Wqkv));
            CGCHK(new_vec(m, 3 * H, &L.bqkv));
            const char* qkvn[3] = {"query", "key", "value"};
            for (int j = 0; j < 3; ++j) {
                add_view(m, lp + "attention.self." + qkvn[j] + ".weight", L.Wqkv, true, H, H, m->Hk, (int64_t)j * H * m->Hk,
                         INIT_NORMAL, S);
                add_view(m, lp + "attention.self." + qkvn[j] + ".bias", L.bqkv, false, 1, H, H, (int64_t)j * H, INIT_ZERO, 0);
            }
            CGCHK(add_mat(m, lp + "attention.output.dense.weight", H, H, S, &L.Wo));
            CGCHK(add_vec(m, lp + "attention.output.dense.bias", H, INIT_ZERO, 0, &L.bo));
            CGCHK(add_vec(m, lp + "attention.output.LayerNorm.weight", H, INIT_ONE, 0, &L.lnw));
            CGCHK(add_vec(m, lp + "attention.output.LayerNorm.bias", H, INIT_ZERO, 0, &L.lnb));
            L.xattn_index = -1;
            if (i % c.qf_xattn_freq == 0) {
                L.xattn_index = xi;
                CGCHK(add_mat(m, lp + "crossattention.self.query.weight", H, H, S, &L.Wxq));
                CGCHK(add_vec(m, lp + "crossattention.self.query.bias", H, INIT_ZERO, 0, &L.bxq));
                add_view(m, lp + "crossattention.self.key.weight", m->Wkv_all, true, H, D, m->Dk,
                         (int64_t)(xi * 2) * H * m->Dk, INIT_NORMAL, S);
                add_view(m, lp + "crossattention.self.key.bias", m->bkv_all, false, 1, H, H, (int64_t)(xi * 2) * H, INIT_ZERO, 0);
                add_view(m, lp + "crossattention.self.value.weight", m->Wkv_all, true, H, D, m->Dk,
                         (int64_t)(xi * 2 + 1) * H * m->Dk, INIT_NORMAL, S);
                add_view(m, lp + "crossattention.self.value.bias", m->bkv_all, false, 1, H, H, (int64_t)(xi * 2 + 1) * H,
                         INIT_ZERO, 0);
                CGCHK(add_mat(m, lp + "crossattention.output.dense.weight", H, H, S, &L.Wxo));
                CGCHK(add_vec(m, lp + "crossattention.output.dense.bias", H, INIT_ZERO, 0, &L.bxo));
                CGCHK(add_vec(m, lp + "crossattention.output.LayerNorm.weight", H, INIT_ONE, 0, &L.xlnw));
                CGCHK(add_vec(m, lp + "crossattention.output.LayerNorm.bias", H, INIT_ZERO, 0, &L.xlnb));
                ++xi;
            }
            CGCHK(add_mat(m, lp + "intermediate_query.dense.weight", m->F, H, S, &L.Wi));
            CGCHK(add_vec(m, lp + "intermediate_query.dense.bias", m->F, INIT_ZERO, 0, &L.bi));
            CGCHK(add_mat(m, lp + "output_query.dense.weight", H, m->F, S, &L.Wo2));
            CGCHK(add_vec(m, lp + "output_query.dense.bias", H, INIT_ZERO, 0, &L.bo2));
            CGCHK(add_vec(m, lp + "output_query.LayerNorm.weight", H, INIT_ONE, 0, &L.ln2w));
            CGCHK(add_vec(m, lp + "output_query.LayerNorm.bias", H, INIT_ZERO, 0, &L.ln2b));
        }
        CGCHK(add_mat(m, "llama_proj.weight", m->PD, H, S, &m->Wproj_l));
        CGCHK(add_vec(m, "llama_proj.bias", m->PD, INIT_ZERO, 0, &m->bproj_l));
        CGCHK(add_mat(m, "head.weight", m->K, m->PD, 0.05f, &m->Whead));
    } else {
        CGCHK(add_mat(m, "head.weight", m->K, D, 0.05f, &m->Whead));
    }
    CGCHK(add_vec(m, "head.bias", m->K, INIT_ZERO, 0, &m->bhead));

    // ---- workspace for max_batch samples
    const int64_t nb = c.max_batch, M = nb * T;
    m->ld_qkv = 3 * D;
    CGCHK(new_act16(m, nb * m->P, m->Kpatch_p, &m->Apatch));
    CGCHK(new_act32(m, M, D, &m->resid));
    CGCHK(new_act16(m, M, D, &m->xn));
    CGCHK(new_act16(m, M, 3 * D, &m->qkv));
    CGCHK(new_act16(m, M, D, &m->attn));
    CGCHK(new_act16(m, M, m->mlp, &m->hid));
    CGCHK(new_act16(m, M, D, &m->delta));
    CGCHK(new_act16(m, M, D, &m->delta2));
    CGCHK(new_act16(m, nb, D, &m->cls16));
    CGCHK(new_act32(m, nb, m->K, &m->logits));
    if (full) {
        const int64_t MQ = nb * m->Q;
        m->ld_kv = (int64_t)m->nx * 2 * H;
        CGCHK(new_act16(m, M, D, &m->emb));
        CGCHK(new_act16(m, M, m->ld_kv, &m->kv_all));
        CGCHK(new_act32(m, m->Q, H, &m->qemb0));
        CGCHK(new_act32(m, MQ, H, &m->qh32));
        CGCHK(new_act16(m, MQ, H, &m->qh16));
        CGCHK(new_act16(m, MQ, 3 * H, &m->qqkv));
        CGCHK(new_act16(m, MQ, H, &m->qctx));
        CGCHK(new_act16(m, MQ, H, &m->qq));
        CGCHK(new_act16(m, MQ, m->F, &m->qff));
        CGCHK(new_act32(m, MQ, H, &m->qtmp));
        CGCHK(new_act32(m, MQ, m->PD, &m->llama));
        CGCHK(new_act16(m, nb, m->PD, &m->pooled));
    }
    return CGPT_OK;
}

// ---- launch helpers -------------------------------------------------------------------------------
cgpt_status gemm(cgpt_model* m, int epi, const half_t* A, int64_t lda, const half_t* W, int64_t ldw, const float* bias,
                 void* out, int64_t ldo, const float* aux, int64_t ldaux, int M, int N, int Kd, int kind, hipStream_t st) {
    GemmParams p;
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias; p.out = out; p.ldo = ldo; p.aux = aux; p.ldaux = ldaux;
    p.M = M; p.N = N; p.K = Kd; p.patches = m->P;
    ProfEvent ev;
    if (m->profile) {
        // timing-only events: no system-scope fence (cache write-back + invalidate) when they are recorded, which would slow down the
        // GEMM that follows -- its A operand is the previous kernel's output, still in L2 / Infinity Cache -- and events come from a pool
        auto take = [&](hipEvent_t* e) -> hipError_t {
            if (!m->event_pool.empty()) { *e = m->event_pool.back(); m->event_pool.pop_back(); return hipSuccess; }
            return hipEventCreateWithFlags(e, hipEventDisableSystemFence);
        };
        HIPCHK(take(&ev.a)); HIPCHK(take(&ev.b));
        HIPCHK(hipEventRecord(ev.a, st));
        if (m->clk_dev && kind >= 0 && kind < 8) p.clk = m->clk_dev + 2 * kind;
    }
    HIPCHK(launch_gemm(epi, p, st));
    if (m->profile) {
        HIPCHK(hipEventRecord(ev.b, st));
        ev.flops = 2.0 * (double)M * (double)N * (double)Kd; ev.kind = kind;
        m->events.push_back(ev);
    }
    return CGPT_OK;
}

cgpt_status attention(const half_t* Q, int64_t ldq, int64_t qbs, const half_t* K, int64_t ldk, const half_t* V, int64_t ldv,
                      int64_t kvbs, half_t* O, int64_t ldo, int64_t obs, int B, int heads, int hd, int Tq, int Tk,
                      hipStream_t st) {
    AttnParams p;
    p.Q = Q; p.ldq = ldq; p.q_batch_stride = qbs; p.K = K; p.ldk = ldk; p.V = V; p.ldv = ldv; p.kv_batch_stride = kvbs;
    p.O = O; p.ldo = ldo; p.o_batch_stride = obs; p.B = B; p.heads = heads; p.head_dim = hd; p.Tq = Tq; p.Tk = Tk;
    p.scale = 1.0f / sqrtf((float)hd);             // eva_vit.py:79 head_dim**-0.5; Qformer.py:244 / sqrt(head_size)
    HIPCHK(launch_attention(p, st));
    return CGPT_OK;
}

// One base-classifier forward for nb samples.  src: the clean image (noise=true) or nb images (noise=false).
// With noise: batch row b is sample first_sample + b for b < na, first_b + (b - na) otherwise.
// llama_out != nullptr (CGPT_MODE_ENCODE_IMG only): llama_proj writes its [nb*Q, proj_dim] fp32 output straight into that
// caller buffer and the build-side head is skipped (cgpt_encode_img*: the consumer is an LLM, not the vote).
cgpt_status forward(cgpt_model* m, const float* src, bool noise, int64_t first_sample, int na, int64_t first_b, int nb,
                    float sigma, uint64_t seed, hipStream_t st, int per = 0, int64_t img_stride = 0, int64_t row0 = 0,
                    float* llama_out = nullptr) {
    const cgpt_config& c = m->cfg;
    const int D = m->D, Dk = m->Dk, T = m->T, P = m->P, M = nb * T;
    m->pending_delta = false;
    if (m->profile && m->batch_log.size() < (1u << 20)) m->batch_log.push_back(nb);
    // K1 + im2col: smoothing.py:95-96 fused into the patch-embed operand (eva_vit.py:202,209)
    if (noise) HIPCHK(launch_noise_im2col(src, c.img_size, c.patch_size, first_sample, na, first_b, nb, sigma, seed, m->Apatch, m->Kpatch_p, st, per, img_stride, row0));
    else HIPCHK(launch_im2col(src, c.img_size, c.patch_size, nb, m->Apatch, m->Kpatch_p, st));
    // patch-embed GEMM + bias + pos_embed, scattered to token rows 1..P; CLS row = cls + pos[0]  (eva_vit.py:333-340)
    CGCHK(gemm(m, EPI_PATCH, m->Apatch, m->Kpatch_p, m->Wpatch, m->Kpatch_p, m->bpatch, m->resid, D, m->pos, D, nb * P, D,
               m->Kpatch_p, 0, st));
    HIPCHK(launch_cls_rows(m->cls, m->pos, m->resid, D, T, nb, D, st));
    const int hd = D / c.vit_heads;
    // Residual adds (x = x + attn(..), x = x + mlp(..), eva_vit.py:180-181) are deferred: proj / fc2 write their fp16
    // output to `delta2` / `delta` (the reference's autocast Linear output is fp16 too) and the LayerNorm kernels apply them
    // while they read the row anyway; no GEMM epilogue reads the fp32 stream.  The stream is WRITTEN once per block: LN1
    // normalises resid + delta (the previous block's fc2 output) without storing the sum, LN2 normalises and stores
    // resid + delta + delta2 (this block's proj output) -- the same fp32 additions in the same order as two stores.
    for (int i = 0; i < c.vit_depth; ++i) {                                   // Block.forward, eva_vit.py:178-185
        const VitLayer& L = m->vit[i];
        HIPCHK(launch_layernorm(m->resid, D, i ? m->delta : nullptr, Dk, L.n1w, L.n1b, c.vit_ln_eps, m->xn, Dk, nullptr, 0, M, D, st,
                                nullptr, 0, /*keep_x=*/1));
        CGCHK(gemm(m, EPI_F16, m->xn, Dk, L.Wqkv, Dk, L.bqkv, m->qkv, m->ld_qkv, nullptr, 0, M, 3 * D, Dk, 2, st));
        CGCHK(attention(m->qkv, m->ld_qkv, (int64_t)T * m->ld_qkv, m->qkv + D, m->ld_qkv, m->qkv + 2 * D, m->ld_qkv,
                        (int64_t)T * m->ld_qkv, m->attn, Dk, (int64_t)T * Dk, nb, c.vit_heads, hd, T, T, st));
        CGCHK(gemm(m, EPI_F16, m->attn, Dk, L.Wproj, Dk, L.bproj, m->delta2, Dk, nullptr, 0, M, D, Dk, 3, st));
        HIPCHK(launch_layernorm(m->resid, D, i ? m->delta : nullptr, Dk, L.n2w, L.n2b, c.vit_ln_eps, m->xn, Dk, nullptr, 0, M, D, st,
                                m->delta2, Dk, 0));
        CGCHK(gemm(m, EPI_F16_GELU, m->xn, Dk, L.Wfc1, Dk, L.bfc1, m->hid, m->mlp_k, nullptr, 0, M, m->mlp, Dk, 1, st));
        CGCHK(gemm(m, EPI_F16, m->hid, m->mlp_k, L.Wfc2, m->mlp_k, L.bfc2, m->delta, Dk, nullptr, 0, M, D, m->mlp_k, 4, st));
    }
    if (c.mode == CGPT_MODE_VIT_HEAD) {
        // ln_vision on the CLS rows only (row stride T*D), applying the last block's pending update to those rows on the way
        // (the head reads nothing else; cgpt_get_activation("vit_out") applies it to the other rows on demand), then the head
        HIPCHK(launch_layernorm(m->resid, (int64_t)T * D, m->delta, (int64_t)T * Dk, m->lnvw, m->lnvb, c.ln_vision_eps, m->cls16, Dk,
                                nullptr, 0, nb, D, st));
        m->pending_delta = true;
        CGCHK(gemm(m, EPI_F32, m->cls16, Dk, m->Whead, Dk, m->bhead, m->logits, m->K, nullptr, 0, nb, m->K, Dk, 0, st));
        m->last_nb = nb;
        return CGPT_OK;
    }
    // ---- MiniGPT4.encode_img tail: ln_vision -> Q-Former -> llama_proj (minigpt4.py:129-141)
    const int H = m->H, Hk = m->Hk, Q = m->Q, MQ = nb * Q, qhd = H / c.qf_heads;
    HIPCHK(launch_layernorm(m->resid, D, m->delta, Dk, m->lnvw, m->lnvb, c.ln_vision_eps, m->emb, Dk, nullptr, 0, M, D, st));
    // K/V of every cross-attention layer in one GEMM over the image tokens (Qformer.py:185-188,203-204)
    CGCHK(gemm(m, EPI_F16, m->emb, Dk, m->Wkv_all, Dk, m->bkv_all, m->kv_all, m->ld_kv, nullptr, 0, M, (int)m->ld_kv, Dk, 0, st));
    // embeddings: LayerNorm(query_tokens) (Qformer.py:104-106), expanded over the batch (minigpt4.py:132)
    HIPCHK(launch_layernorm(m->qtok, H, nullptr, 0, m->qembw, m->qembb, c.qf_ln_eps, nullptr, 0, m->qemb0, H, Q, H, st));
    HIPCHK(launch_broadcast_rows(m->qemb0, Q, H, nb, m->qh32, H, m->qh16, Hk, st));
    for (int i = 0; i < c.qf_layers; ++i) {                                   // BertLayer.forward, Qformer.py:402-474
        const QfLayer& L = m->qf[i];
        CGCHK(gemm(m, EPI_F16, m->qh16, Hk, L.Wqkv, Hk, L.bqkv, m->qqkv, 3 * H, nullptr, 0, MQ, 3 * H, Hk, 0, st));
        CGCHK(attention(m->qqkv, 3 * H, (int64_t)Q * 3 * H, m->qqkv + H, 3 * H, m->qqkv + 2 * H, 3 * H, (int64_t)Q * 3 * H,
                        m->qctx, Hk, (int64_t)Q * Hk, nb, c.qf_heads, qhd, Q, Q, st));
        CGCHK(gemm(m, EPI_RESID, m->qctx, Hk, L.Wo, Hk, L.bo, m->qtmp, H, m->qh32, H, MQ, H, Hk, 0, st));   // Qformer.py:285-288
        HIPCHK(launch_layernorm(m->qtmp, H, nullptr, 0, L.lnw, L.lnb, c.qf_ln_eps, m->qh16, Hk, m->qh32, H, MQ, H, st));
        if (L.xattn_index >= 0) {                                            // Qformer.py:432-447
            const half_t* Kx = m->kv_all + (int64_t)L.xattn_index * 2 * H;
            CGCHK(gemm(m, EPI_F16, m->qh16, Hk, L.Wxq, Hk, L.bxq, m->qq, Hk, nullptr, 0, MQ, H, Hk, 0, st));
            CGCHK(attention(m->qq, Hk, (int64_t)Q * Hk, Kx, m->ld_kv, Kx + H, m->ld_kv, (int64_t)T * m->ld_kv, m->qctx, Hk,
                            (int64_t)Q * Hk, nb, c.qf_heads, qhd, Q, T, st));
            CGCHK(gemm(m, EPI_RESID, m->qctx, Hk, L.Wxo, Hk, L.bxo, m->qtmp, H, m->qh32, H, MQ, H, Hk, 0, st));
            HIPCHK(launch_layernorm(m->qtmp, H, nullptr, 0, L.xlnw, L.xlnb, c.qf_ln_eps, m->qh16, Hk, m->qh32, H, MQ, H, st));
        }
        // feed_forward_chunk_query, Qformer.py:481-484
        CGCHK(gemm(m, EPI_F16_GELU, m->qh16, Hk, L.Wi, Hk, L.bi, m->qff, m->Fk, nullptr, 0, MQ, m->F, Hk, 0, st));
        CGCHK(gemm(m, EPI_RESID, m->qff, m->Fk, L.Wo2, m->Fk, L.bo2, m->qtmp, H, m->qh32, H, MQ, H, m->Fk, 0, st));
        HIPCHK(launch_layernorm(m->qtmp, H, nullptr, 0, L.ln2w, L.ln2b, c.qf_ln_eps, m->qh16, Hk, m->qh32, H, MQ, H, st));
    }
    if (llama_out) {                                                          // inputs_llama, minigpt4.py:141
        CGCHK(gemm(m, EPI_F32, m->qh16, Hk, m->Wproj_l, Hk, m->bproj_l, llama_out, m->PD, nullptr, 0, MQ, m->PD, Hk, 0, st));
        m->last_nb = 0;                                                       // the workspace no longer holds a complete forward
        return CGPT_OK;
    }
    CGCHK(gemm(m, EPI_F32, m->qh16, Hk, m->Wproj_l, Hk, m->bproj_l, m->llama, m->PD, nullptr, 0, MQ, m->PD, Hk, 0, st));
    // build-side label head on the mean of the query tokens
    HIPCHK(launch_mean_rows(m->llama, m->PD, Q, m->PD, nb, m->pooled, m->PDk, st));
    CGCHK(gemm(m, EPI_F32, m->pooled, m->PDk, m->Whead, m->PDk, m->bhead, m->logits, m->K, nullptr, 0, nb, m->K, m->PDk, 0, st));
    m->last_nb = nb;
    return CGPT_OK;
}

const View* find_view(cgpt_model* m, const char* name) {
    if (!m || !name) return nullptr;
    auto it = m->index.find(name);
    return it == m->index.end() ? nullptr : &m->views[it->second];
}

}  // namespace

// =========================================================================================== C-ABI
extern "C" {

const char* cgpt_last_error(void) { return g_last_error.c_str(); }
const char* cgpt_version(void) { return "cgpt 0.1 (gfx950)"; }

cgpt_status cgpt_create(const cgpt_config* cfg, cgpt_handle* out) {
    if (!cfg || !out) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(cgpt_config))
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_create: cgpt_config.struct_size mismatch (header/library drift)");
    const cgpt_config& c = *cfg;
    if (c.mode != CGPT_MODE_VIT_HEAD && c.mode != CGPT_MODE_ENCODE_IMG) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_create: bad mode");
    if (c.num_classes < 1 || c.max_batch < 1 || c.img_size < 1 || c.patch_size < 1 || c.img_size % c.patch_size ||
        (c.img_size & 3) || c.vit_dim < 8 || (c.vit_dim & 7) || c.vit_depth < 1 || c.vit_heads < 1 || c.vit_dim % c.vit_heads ||
        c.vit_mlp < 8 || (c.vit_mlp & 7))
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_create: bad ViT dimensions");
    const int hd = c.vit_dim / c.vit_heads;
    if (hd != 88 && hd != 64) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_create: ViT head_dim must be 88 or 64");
    const int T = (c.img_size / c.patch_size) * (c.img_size / c.patch_size) + 1;
    if (T > 4097) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_create: more than 4097 tokens per image");
    if (c.mode == CGPT_MODE_ENCODE_IMG) {
        if (c.qf_layers < 1 || c.qf_dim < 8 || c.qf_heads < 1 || c.qf_dim % c.qf_heads || c.qf_dim / c.qf_heads != 64 ||
            c.qf_ffn < 8 || (c.qf_ffn & 7) || c.qf_queries < 1 || c.qf_queries > 32 || c.qf_xattn_freq < 1 || c.proj_dim < 8 ||
            (c.proj_dim & 7))
            return cgpt_fail(CGPT_ERR_INVALID, "cgpt_create: bad Q-Former dimensions (head_dim 64, <= 32 queries)");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return cgpt_fail(CGPT_ERR_NO_DEVICE, "cgpt_create: no HIP device visible (this library has no CPU fallback)");
    if (c.device < 0 || c.device >= ndev) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_create: bad device ordinal");
    HIPCHK(hipSetDevice(c.device));
    cgpt_model* m = new cgpt_model();
    m->cfg = c;
    cgpt_status s = build(m);
    if (s == CGPT_OK) s = dev_alloc(m, 16 * sizeof(unsigned long long), (void**)&m->clk_dev);
    if (s != CGPT_OK) { cgpt_destroy(m); return s; }
    HIPCHK(hipDeviceSynchronize());
    *out = m;
    return CGPT_OK;
}

cgpt_status cgpt_destroy(cgpt_handle h) {
    if (!h) return CGPT_OK;
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    for (auto& e : h->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : h->event_pool) (void)hipEventDestroy(e);
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
    return CGPT_OK;
}

int64_t cgpt_weight_numel(cgpt_handle h, const char* name) {
    const View* v = find_view(h, name);
    return v ? v->rows * v->cols : -1;
}
const char* cgpt_weight_name(cgpt_handle h, int32_t index) {
    if (!h || index < 0 || index >= (int32_t)h->views.size()) return nullptr;
    return h->views[index].name.c_str();
}

cgpt_status cgpt_load_weight(cgpt_handle h, const char* name, const float* data_host, int64_t numel) {
    if (!h || !name || !data_host) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_load_weight: null argument");
    const View* v = find_view(h, name);
    if (!v) return cgpt_fail(CGPT_ERR_NOT_FOUND, std::string("cgpt_load_weight: unknown weight '") + name + "'");
    if (numel != v->rows * v->cols)
        return cgpt_fail(CGPT_ERR_INVALID, std::string("cgpt_load_weight: '") + name + "' expects " +
                                               std::to_string(v->rows * v->cols) + " elements, got " + std::to_string(numel));
    HIPCHK(hipSetDevice(h->cfg.device));
    if (v->f16) {
        float* tmp = nullptr;
        HIPCHK(hipMalloc(&tmp, (size_t)numel * sizeof(float)));
        hipError_t e = hipMemcpy(tmp, data_host, (size_t)numel * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = launch_f32_to_f16(tmp, v->cols, (half_t*)v->base + v->off, v->ld, v->rows, v->cols, 0);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        (void)hipFree(tmp);
        HIPCHK(e);
    } else {
        HIPCHK(hipMemcpy2D((float*)v->base + v->off, (size_t)v->ld * sizeof(float), data_host, (size_t)v->cols * sizeof(float),
                           (size_t)v->cols * sizeof(float), (size_t)v->rows, hipMemcpyHostToDevice));
    }
    return CGPT_OK;
}

cgpt_status cgpt_get_weight(cgpt_handle h, const char* name, float* out_host, int64_t numel) {
    if (!h || !name || !out_host) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_get_weight: null argument");
    const View* v = find_view(h, name);
    if (!v) return cgpt_fail(CGPT_ERR_NOT_FOUND, std::string("cgpt_get_weight: unknown weight '") + name + "'");
    if (numel != v->rows * v->cols) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_get_weight: numel mismatch");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    if (v->f16) {
        float* tmp = nullptr;
        HIPCHK(hipMalloc(&tmp, (size_t)numel * sizeof(float)));
        hipError_t e = launch_f16_to_f32((const half_t*)v->base + v->off, v->ld, tmp, v->cols, v->rows, v->cols, 0);
        if (e == hipSuccess) e = hipMemcpy(out_host, tmp, (size_t)numel * sizeof(float), hipMemcpyDeviceToHost);
        (void)hipFree(tmp);
        HIPCHK(e);
    } else {
        HIPCHK(hipMemcpy2D(out_host, (size_t)v->cols * sizeof(float), (const float*)v->base + v->off,
                           (size_t)v->ld * sizeof(float), (size_t)v->cols * sizeof(float), (size_t)v->rows, hipMemcpyDeviceToHost));
    }
    return CGPT_OK;
}

cgpt_status cgpt_init_synthetic_weights(cgpt_handle h, uint64_t seed, void* stream) {
    if (!h) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_init_synthetic_weights: null handle");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->cfg.device));
    for (size_t i = 0; i < h->views.size(); ++i) {
        const View& v = h->views[i];
        if (v.init == INIT_NORMAL) {
            void* dst = v.f16 ? (void*)((half_t*)v.base + v.off) : (void*)((float*)v.base + v.off);
            HIPCHK(launch_fill_normal(dst, v.f16 ? 1 : 0, v.rows, v.cols, v.ld, 0.0f, v.init_std, seed, (uint64_t)i, st));
        } else {
            // 1-D fp32 parameters (rows == 1): LayerNorm weight 1, every bias 0 (eva_vit.py:316-323)
            HIPCHK(launch_fill_const((float*)v.base + v.off, v.cols, v.init == INIT_ONE ? 1.0f : 0.0f, st));
        }
    }
    return CGPT_OK;
}

static cgpt_status check_call(cgpt_handle h, const void* p1, const void* p2, int64_t num, const char* who) {
    if (!h || !p1 || !p2) return cgpt_fail(CGPT_ERR_INVALID, std::string(who) + ": null argument");
    if (num < 0) return cgpt_fail(CGPT_ERR_INVALID, std::string(who) + ": negative sample count");
    return CGPT_OK;
}

// cgpt_set_option("sync_batches", 1): wait for the stream after every classifier batch of the sample_counts* entry points.  Measurement aid:
// a profiler that keeps one record per dispatch in flight (rocprofv3 --pmc) runs out of room when one call enqueues tens of thousands of
// dispatches without a host synchronisation (40 batches x ~400 kernels at 51 images per cgpt_sample_counts_images call).  Never changes a result.
static int g_sync_batches = 0;
// cgpt_set_option("trace_batches", 1): one stderr line per classifier batch ENQUEUED by the sample_counts* entry points (host side, in
// front of the optional synchronisation): tells a log how far a call got (profiles/r06/pmc_sigsegv.txt).  Measurement aid.
static int g_trace_batches = 0;
static long g_batches_enqueued = 0;
#define CGPT_BATCH_DONE(st) do { \
        ++g_batches_enqueued; \
        if (g_trace_batches) fprintf(stderr, "libcgpt: classifier batch %ld enqueued%s\n", g_batches_enqueued, g_sync_batches ? " (synchronising)" : ""); \
        if (g_sync_batches) HIPCHK(hipStreamSynchronize(st)); } while (0)

cgpt_status cgpt_sample_counts(cgpt_handle h, const float* x_dev, int64_t first_sample, int64_t num, int64_t batch_size,
                               float sigma, uint64_t noise_seed, int64_t* counts_dev, void* stream) {
    CGCHK(check_call(h, x_dev, counts_dev, num, "cgpt_sample_counts"));
    if (batch_size < 1 || batch_size > h->cfg.max_batch)
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_sample_counts: batch_size must be in [1, max_batch]");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->cfg.device));
    int64_t done = 0;
    while (done < num) {                                   // smoothing.py:91-98
        const int nb = (int)((num - done < batch_size) ? (num - done) : batch_size);
        CGCHK(forward(h, x_dev, true, first_sample + done, nb, 0, nb, sigma, noise_seed, st));
        HIPCHK(launch_vote(h->logits, h->K, nb, h->K, counts_dev, nb, counts_dev, st));
        CGPT_BATCH_DONE(st);
        done += nb;
    }
    return CGPT_OK;
}

cgpt_status cgpt_sample_counts2(cgpt_handle h, const float* x_dev, int64_t first_a, int64_t num_a, int64_t* counts_a_dev,
                                int64_t first_b, int64_t num_b, int64_t* counts_b_dev, int64_t batch_size, float sigma,
                                uint64_t noise_seed, void* stream) {
    CGCHK(check_call(h, x_dev, counts_a_dev, num_a, "cgpt_sample_counts2"));
    if (!counts_b_dev || num_b < 0) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_sample_counts2: bad second range");
    if (batch_size < 1 || batch_size > h->cfg.max_batch)
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_sample_counts2: batch_size must be in [1, max_batch]");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->cfg.device));
    const int64_t total = num_a + num_b;
    int64_t done = 0;                                      // position in the concatenation of the two ranges
    while (done < total) {
        const int nb = (int)((total - done < batch_size) ? (total - done) : batch_size);
        const int na = (int)(done >= num_a ? 0 : ((num_a - done < nb) ? (num_a - done) : nb));   // rows of range A in this batch
        const int64_t fa = first_a + done;                                                     // used only when na > 0
        const int64_t fb = first_b + (done > num_a ? done - num_a : 0);
        CGCHK(forward(h, x_dev, true, fa, na, fb, nb, sigma, noise_seed, st));
        HIPCHK(launch_vote(h->logits, h->K, nb, h->K, counts_a_dev, na, counts_b_dev, st));
        CGPT_BATCH_DONE(st);
        done += nb;
    }
    return CGPT_OK;
}

cgpt_status cgpt_sample_counts_images(cgpt_handle h, const float* x_dev, int64_t num_images, int64_t first_a, int64_t num_a,
                                      int64_t first_b, int64_t num_b, int64_t image_stride, int64_t* counts_dev, float sigma,
                                      uint64_t noise_seed, void* stream) {
    CGCHK(check_call(h, x_dev, counts_dev, num_images, "cgpt_sample_counts_images"));
    const int64_t per = num_a + num_b;
    if (num_a < 0 || num_b < 0 || per < 1 || per > 0x7fffffff)
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_sample_counts_images: need num_a, num_b >= 0 and num_a + num_b >= 1");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->cfg.device));
    // the rows of all images form one sequence (image-major); batches are windows of max_batch rows of it, not aligned to images
    const int64_t total = num_images * per;
    for (int64_t r0 = 0; r0 < total; r0 += h->cfg.max_batch) {
        const int nb = (int)((total - r0 < h->cfg.max_batch) ? (total - r0) : h->cfg.max_batch);
        CGCHK(forward(h, x_dev, true, first_a, (int)num_a, first_b, nb, sigma, noise_seed, st, (int)per, image_stride, r0));
        HIPCHK(launch_vote(h->logits, h->K, nb, h->K, counts_dev, num_a, counts_dev + h->K, st, (int)per, r0));
        CGPT_BATCH_DONE(st);
    }
    return CGPT_OK;
}

cgpt_status cgpt_forward_logits(cgpt_handle h, const float* x_dev, int64_t first_sample, int64_t num, float sigma,
                                uint64_t noise_seed, float* logits_dev, void* stream) {
    CGCHK(check_call(h, x_dev, logits_dev, num, "cgpt_forward_logits"));
    if (num < 1 || num > h->cfg.max_batch) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_forward_logits: num must be in [1, max_batch]");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->cfg.device));
    CGCHK(forward(h, x_dev, true, first_sample, (int)num, 0, (int)num, sigma, noise_seed, st));
    HIPCHK(hipMemcpyAsync(logits_dev, h->logits, (size_t)num * h->K * sizeof(float), hipMemcpyDeviceToDevice, st));
    return CGPT_OK;
}

cgpt_status cgpt_classify(cgpt_handle h, const float* images_dev, int64_t num, float* logits_dev, void* stream) {
    CGCHK(check_call(h, images_dev, logits_dev, num, "cgpt_classify"));
    if (num < 1 || num > h->cfg.max_batch) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_classify: num must be in [1, max_batch]");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->cfg.device));
    CGCHK(forward(h, images_dev, false, 0, (int)num, 0, (int)num, 0.f, 0, st));
    HIPCHK(hipMemcpyAsync(logits_dev, h->logits, (size_t)num * h->K * sizeof(float), hipMemcpyDeviceToDevice, st));
    return CGPT_OK;
}

static cgpt_status encode_check(cgpt_handle h, const void* src, const void* out, int64_t num, const char* who) {
    CGCHK(check_call(h, src, out, num, who));
    if (h->cfg.mode != CGPT_MODE_ENCODE_IMG)
        return cgpt_fail(CGPT_ERR_STATE, std::string(who) + ": the handle was not created with CGPT_MODE_ENCODE_IMG");
    return CGPT_OK;
}

cgpt_status cgpt_encode_img(cgpt_handle h, const float* images_dev, int64_t num, float* inputs_llama_dev, void* stream) {
    CGCHK(encode_check(h, images_dev, inputs_llama_dev, num, "cgpt_encode_img"));
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->cfg.device));
    const int64_t per = (int64_t)h->Q * h->PD, chw = (int64_t)3 * h->cfg.img_size * h->cfg.img_size;
    for (int64_t done = 0; done < num; done += h->cfg.max_batch) {
        const int nb = (int)((num - done < h->cfg.max_batch) ? (num - done) : h->cfg.max_batch);
        CGCHK(forward(h, images_dev + done * chw, false, 0, nb, 0, nb, 0.f, 0, st, 0, 0, 0, inputs_llama_dev + done * per));
    }
    return CGPT_OK;
}

cgpt_status cgpt_encode_img_noisy(cgpt_handle h, const float* x_dev, int64_t first_sample, int64_t num, float sigma,
                                  uint64_t noise_seed, float* inputs_llama_dev, void* stream) {
    CGCHK(encode_check(h, x_dev, inputs_llama_dev, num, "cgpt_encode_img_noisy"));
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->cfg.device));
    const int64_t per = (int64_t)h->Q * h->PD;
    for (int64_t done = 0; done < num; done += h->cfg.max_batch) {
        const int nb = (int)((num - done < h->cfg.max_batch) ? (num - done) : h->cfg.max_batch);
        CGCHK(forward(h, x_dev, true, first_sample + done, nb, 0, nb, sigma, noise_seed, st, 0, 0, 0, inputs_llama_dev + done * per));
    }
    return CGPT_OK;
}

cgpt_status cgpt_get_activation(cgpt_handle h, const char* what, float* out_dev, int64_t numel, void* stream) {
    if (!h || !what || !out_dev) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_get_activation: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int nb = h->last_nb;
    if (nb <= 0) return cgpt_fail(CGPT_ERR_STATE, "cgpt_get_activation: no forward has run");
    const std::string w(what);
    const bool full = h->cfg.mode == CGPT_MODE_ENCODE_IMG;
    HIPCHK(hipSetDevice(h->cfg.device));
    if (w == "vit_out") {
        if (numel != (int64_t)nb * h->T * h->D) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_get_activation: numel mismatch");
        if (h->pending_delta) {
            HIPCHK(launch_add_delta(h->resid, h->D, h->delta, h->Dk, (int64_t)nb * h->T, h->D, st, h->T));
            h->pending_delta = false;
        }
        HIPCHK(hipMemcpyAsync(out_dev, h->resid, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else if (w == "ln_vision" && full) {
        if (numel != (int64_t)nb * h->T * h->D) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_get_activation: numel mismatch");
        HIPCHK(launch_f16_to_f32(h->emb, h->Dk, out_dev, h->D, (int64_t)nb * h->T, h->D, st));
    } else if (w == "qformer" && full) {
        if (numel != (int64_t)nb * h->Q * h->H) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_get_activation: numel mismatch");
        HIPCHK(hipMemcpyAsync(out_dev, h->qh32, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else if (w == "llama" && full) {
        if (numel != (int64_t)nb * h->Q * h->PD) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_get_activation: numel mismatch");
        HIPCHK(hipMemcpyAsync(out_dev, h->llama, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        return cgpt_fail(CGPT_ERR_NOT_FOUND, "cgpt_get_activation: unknown activation for this mode");
    }
    return CGPT_OK;
}

cgpt_status cgpt_noise_batch(const float* x_dev, int64_t chw, int64_t first_sample, int64_t num, float sigma,
                             uint64_t noise_seed, float* out_dev, void* stream) {
    if (!x_dev || !out_dev || chw < 1 || num < 0) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_noise_batch: bad argument");
    HIPCHK(launch_noise_batch(x_dev, chw, first_sample, num, sigma, noise_seed, out_dev, (hipStream_t)stream));
    return CGPT_OK;
}

cgpt_status cgpt_rgf_step(const float* x_adv_dev, const float* x_clean_dev, int64_t chw, int64_t first_dir, int32_t num_dirs,
                          const float* coeffs_host, float lr, float eps, uint64_t noise_seed, float* out_dev, void* stream) {
    if (!x_adv_dev || !x_clean_dev || !out_dev || !coeffs_host || chw < 1 || num_dirs < 1 || num_dirs > CGPT_RGF_MAX_DIRS || !(eps >= 0.f))
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_rgf_step: bad argument (1 <= num_dirs <= 32)");
    HIPCHK(launch_rgf_step(x_adv_dev, x_clean_dev, chw, first_dir, num_dirs, coeffs_host, lr, eps, noise_seed, out_dev, (hipStream_t)stream));
    return CGPT_OK;
}

cgpt_status cgpt_vote(const float* logits_dev, int64_t num, int32_t num_classes, int64_t* counts_dev, void* stream) {
    if (!logits_dev || !counts_dev || num < 0 || num_classes < 1) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_vote: bad argument");
    HIPCHK(launch_vote(logits_dev, num_classes, num, num_classes, counts_dev, num, counts_dev, (hipStream_t)stream));
    return CGPT_OK;
}

// The one collective of the path: sum the int64 vote histograms of all ranks in place.  RCCL is not linked into
// this library.  A communicator is only valid inside the RCCL instance that created it, so ncclAllReduce is taken from a library
// that is ALREADY MAPPED in the process and never from a freshly loaded one: the caller either hands the function over
// (cgpt_allreduce_counts_fn) or exactly one librccl must be mapped, whatever its dlopen flags (torch's bundled copy is
// RTLD_LOCAL: dlsym(RTLD_DEFAULT) does not see it, dl_iterate_phdr + dlopen(RTLD_NOLOAD) does).  Nothing is cached: no shared
// state, any thread may call.  ncclInt64 = 4, ncclSum = 0 (rccl.h).
typedef int (*cgpt_nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);

static int collect_rccl(struct dl_phdr_info* info, size_t, void* data) {
    const char* name = info->dlpi_name;
    if (name && *name) {
        const char* base = strrchr(name, '/');
        base = base ? base + 1 : name;
        if (strncmp(base, "librccl.so", 10) == 0) static_cast<std::vector<std::string>*>(data)->push_back(name);
    }
    return 0;
}

static cgpt_status allreduce_counts_call(const char* who, void* nccl_allreduce, void* rccl_comm, int64_t* counts_dev, int64_t count,
                                         void* stream) {
    const int rc = ((cgpt_nccl_allreduce_fn)nccl_allreduce)(counts_dev, counts_dev, (size_t)count, /*ncclInt64*/ 4, /*ncclSum*/ 0,
                                                            rccl_comm, (hipStream_t)stream);
    if (rc != 0) return cgpt_fail(CGPT_ERR_HIP, std::string(who) + ": ncclAllReduce returned " + std::to_string(rc));
    return CGPT_OK;
}

cgpt_status cgpt_allreduce_counts_fn(void* nccl_allreduce, void* rccl_comm, int64_t* counts_dev, int64_t count, void* stream) {
    if (!nccl_allreduce || !rccl_comm || !counts_dev || count < 1)
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_allreduce_counts_fn: bad argument");
    return allreduce_counts_call("cgpt_allreduce_counts_fn", nccl_allreduce, rccl_comm, counts_dev, count, stream);
}

cgpt_status cgpt_allreduce_counts(void* rccl_comm, int64_t* counts_dev, int64_t count, void* stream) {
    if (!rccl_comm || !counts_dev || count < 1) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_allreduce_counts: bad argument");
    std::vector<std::string> mapped;
    dl_iterate_phdr(collect_rccl, &mapped);
    if (mapped.empty())
        return cgpt_fail(CGPT_ERR_STATE, "cgpt_allreduce_counts: no librccl is mapped in this process, so rccl_comm cannot be a live "
                                         "communicator (create it with the RCCL you loaded, then call again)");
    if (mapped.size() > 1)
        return cgpt_fail(CGPT_ERR_STATE, "cgpt_allreduce_counts: " + std::to_string(mapped.size()) + " RCCL instances are mapped (" +
                                         mapped[0] + ", " + mapped[1] + "): pass the ncclAllReduce of the one that created the "
                                         "communicator to cgpt_allreduce_counts_fn");
    void* lib = dlopen(mapped[0].c_str(), RTLD_NOLOAD | RTLD_NOW);          // a handle to the mapped instance; loads nothing
    void* fn = lib ? dlsym(lib, "ncclAllReduce") : nullptr;
    if (lib) dlclose(lib);                                                   // drops only the reference RTLD_NOLOAD took
    if (!fn) return cgpt_fail(CGPT_ERR_STATE, "cgpt_allreduce_counts: " + mapped[0] + " is mapped but its ncclAllReduce cannot be reached");
    return allreduce_counts_call("cgpt_allreduce_counts", fn, rccl_comm, counts_dev, count, stream);
}

cgpt_status cgpt_certify_device(const int64_t* counts_selection_dev, const int64_t* counts_estimation_dev, int32_t num_classes,
                                int64_t n, double alpha, double sigma, double* out2_dev, void* stream) {
    if (!counts_selection_dev || !counts_estimation_dev || !out2_dev || num_classes < 1 || n < 1 || !(alpha > 0.0 && alpha < 1.0))
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_certify_device: bad argument");
    HIPCHK(launch_finalize(counts_selection_dev, counts_estimation_dev, num_classes, n, alpha, sigma, 0, out2_dev, (hipStream_t)stream));
    return CGPT_OK;
}

cgpt_status cgpt_predict_device(const int64_t* counts_dev, int32_t num_classes, double alpha, double* out2_dev, void* stream) {
    if (!counts_dev || !out2_dev || num_classes < 2 || !(alpha > 0.0 && alpha < 1.0))
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_predict_device: bad argument (num_classes >= 2 required)");
    HIPCHK(launch_finalize(counts_dev, counts_dev, num_classes, 1, alpha, 1.0, 1, out2_dev, (hipStream_t)stream));
    return CGPT_OK;
}

cgpt_status cgpt_set_option(const char* key, int32_t value) {
    if (!key) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_set_option: null key");
    const std::string k(key);
    if (k == "gemm_kernel") {
#ifdef CGPT_LAB
        const bool ok = value >= 0 && value <= 15;
#else
        const bool ok = value == 0 || value == 1 || value == 3 || value == 4 || value == 14;
#endif
        if (!ok) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_set_option: gemm_kernel must be 0, 1, 3, 4 or 14");
        g_gemm_kernel = value;
        return CGPT_OK;
    }
    if (k == "gemm_ablate") {
#ifndef CGPT_LAB
        if (value & ~(512 | 16384))
            return cgpt_fail(CGPT_ERR_INVALID, "cgpt_set_option: gemm_ablate (test-only) accepts the result-preserving bits 512|16384");
#endif
        g_gemm_ablate = value;
        return CGPT_OK;
    }
    if (k == "sync_batches") { g_sync_batches = value != 0; return CGPT_OK; }
    if (k == "trace_batches") { g_trace_batches = value != 0; return CGPT_OK; }
    if (k == "gemm_grid") {
        if (value < 0 || (value & 7)) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_set_option: gemm_grid must be 0 (one workgroup per CU) or a positive multiple of 8");
        g_gemm_grid = value;
        return CGPT_OK;
    }
#ifdef CGPT_LAB
    if (k == "gemm_group_m") { if (value < 1) return cgpt_fail(CGPT_ERR_INVALID, "gemm_group_m >= 1"); g_gemm_group_m = value; return CGPT_OK; }
#endif
    return cgpt_fail(CGPT_ERR_NOT_FOUND, "cgpt_set_option: unknown option '" + k + "'");
}

#ifdef CGPT_LAB
// lab builds only: device buffer receiving per-wave cycle sums of the GEMM (not part of include/cgpt.h)
cgpt_status cgpt_debug_set_gemm_stamps(void* dev_buf) { g_gemm_dbg = (unsigned long long*)dev_buf; return CGPT_OK; }
#endif

cgpt_status cgpt_profile_enable(cgpt_handle h, int32_t on) {
    if (!h) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_profile_enable: null handle");
    if (on && !h->profile) h->batch_log.clear();
    h->profile = on != 0;
    return CGPT_OK;
}

cgpt_status cgpt_profile_batches(cgpt_handle h, int32_t* samples_out, int64_t capacity, int64_t* count_out) {
    if (!h || !count_out || capacity < 0 || (capacity > 0 && !samples_out))
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_profile_batches: null argument");
    const int64_t n = (int64_t)h->batch_log.size();
    *count_out = n;
    for (int64_t i = 0; i < n && i < capacity; ++i) samples_out[i] = h->batch_log[i];
    return CGPT_OK;
}

cgpt_status cgpt_profile_clock(cgpt_handle h, int32_t kind, double* clock_ghz) {
    if (!h || !clock_ghz || kind < 0 || kind >= 8) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_profile_clock: null argument or kind outside 0..7");
    *clock_ghz = 0.0;
    if (!h->clk_dev) return CGPT_OK;
    unsigned long long host[16];
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(host, h->clk_dev, sizeof(host), hipMemcpyDeviceToHost));
    double cyc = 0, ticks = 0;
    for (int k = 0; k < 8; ++k)
        if (kind == 0 || k == kind) { cyc += (double)host[2 * k]; ticks += (double)host[2 * k + 1]; }
    if (ticks > 0) *clock_ghz = cyc / ticks * 0.1;          // s_memrealtime counts at 100 MHz
    return CGPT_OK;
}

cgpt_status cgpt_profile_read(cgpt_handle h, int32_t kind, double* total_ms, double* total_flops, int64_t* launches) {
    if (!h || !total_ms || !total_flops || !launches) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_profile_read: null argument");
    double ms = 0, fl = 0; int64_t n = 0;
    for (auto& e : h->events) {
        HIPCHK(hipEventSynchronize(e.b));
        if (kind == 0 || e.kind == kind) {
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, e.a, e.b));
            ms += t; fl += e.flops; ++n;
        }
    }
    if (kind == 0) {   // reading "all" drains the log (and the in-kernel clock sums: every event has been synchronised above)
        for (auto& e : h->events) { h->event_pool.push_back(e.a); h->event_pool.push_back(e.b); }
        h->events.clear();
        if (h->clk_dev) HIPCHK(hipMemset(h->clk_dev, 0, 16 * sizeof(unsigned long long)));
    }
    *total_ms = ms; *total_flops = fl; *launches = n;
    return CGPT_OK;
}

// ---- raw kernels -----------------------------------------------------------------------------------
cgpt_status cgpt_gemm_f16(const void* A_dev, int64_t lda, const void* W_dev, int64_t ldw, const float* bias_dev,
                          float* C_dev, int64_t ldc, int64_t M, int64_t N, int64_t K, void* stream) {
    if (!A_dev || !W_dev || !C_dev || M < 1 || N < 1 || K < 64 || (K % 64) || (lda % 8) || (ldw % 8))
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_gemm_f16: bad argument (K % 64 == 0, lda/ldw % 8 == 0)");
    GemmParams p;
    p.A = (const half_t*)A_dev; p.lda = lda; p.W = (const half_t*)W_dev; p.ldw = ldw; p.bias = bias_dev;
    p.out = C_dev; p.ldo = ldc; p.aux = nullptr; p.ldaux = 0; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.patches = 1;
    HIPCHK(launch_gemm(EPI_F32, p, (hipStream_t)stream));
    return CGPT_OK;
}

cgpt_status cgpt_linear_f16(const void* A_dev, int64_t lda, const void* W_dev, int64_t ldw, const float* bias_dev,
                            void* out_dev, int64_t ldo, const float* aux_dev, int64_t ldaux, int64_t M, int64_t N, int64_t K,
                            int32_t epilogue, void* stream) {
    if (!A_dev || !W_dev || !out_dev || M < 1 || N < 1 || K < 64 || (K % 64) || (lda % 8) || (ldw % 8) || epilogue < 0 ||
        epilogue > 3 || (epilogue == 3 && !aux_dev))
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_linear_f16: bad argument");
    GemmParams p;
    p.A = (const half_t*)A_dev; p.lda = lda; p.W = (const half_t*)W_dev; p.ldw = ldw; p.bias = bias_dev;
    p.out = out_dev; p.ldo = ldo; p.aux = aux_dev; p.ldaux = ldaux; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.patches = 1;
    HIPCHK(launch_gemm(epilogue, p, (hipStream_t)stream));
    return CGPT_OK;
}

cgpt_status cgpt_attention_f16(const void* Q_dev, int64_t ldq, const void* K_dev, const void* V_dev, int64_t ldkv,
                               void* O_dev, int64_t ldo, int32_t B, int32_t heads, int32_t head_dim, int32_t Tq, int32_t Tk,
                               float scale, void* stream) {
    if (!Q_dev || !K_dev || !V_dev || !O_dev) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_attention_f16: null argument");
    AttnParams p;
    p.Q = (const half_t*)Q_dev; p.ldq = ldq; p.q_batch_stride = (int64_t)Tq * ldq;
    p.K = (const half_t*)K_dev; p.ldk = ldkv; p.V = (const half_t*)V_dev; p.ldv = ldkv; p.kv_batch_stride = (int64_t)Tk * ldkv;
    p.O = (half_t*)O_dev; p.ldo = ldo; p.o_batch_stride = (int64_t)Tq * ldo;
    p.B = B; p.heads = heads; p.head_dim = head_dim; p.Tq = Tq; p.Tk = Tk; p.scale = scale;
    hipError_t e = launch_attention(p, (hipStream_t)stream);
    if (e != hipSuccess) return cgpt_fail(e == hipErrorInvalidValue ? CGPT_ERR_INVALID : CGPT_ERR_HIP,
                                          std::string("cgpt_attention_f16: ") + hipGetErrorString(e));
    return CGPT_OK;
}

cgpt_status cgpt_mfma_sustained(double seconds, double* tflops_out, double* clock_ghz_out) {
    if (!tflops_out || !clock_ghz_out || !(seconds > 0.0 && seconds <= 30.0))
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_mfma_sustained: seconds must be in (0, 30], outputs non-null");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return cgpt_fail(CGPT_ERR_NO_DEVICE, "cgpt_mfma_sustained: no HIP device");
    hipError_t e = run_mfma_sustained(seconds, tflops_out, clock_ghz_out);
    if (e != hipSuccess) return cgpt_fail(CGPT_ERR_HIP, std::string("cgpt_mfma_sustained: ") + hipGetErrorString(e));
    return CGPT_OK;
}

cgpt_status cgpt_layernorm(const float* x_dev, int64_t ldx, const float* gamma_dev, const float* beta_dev, float eps,
                           void* y_dev, int64_t ldy, float* y32_dev, int64_t ldy32, int64_t rows, int32_t D, void* stream) {
    if (!x_dev || !gamma_dev || !beta_dev || (!y_dev && !y32_dev)) return cgpt_fail(CGPT_ERR_INVALID, "cgpt_layernorm: null argument");
    hipError_t e = launch_layernorm(const_cast<float*>(x_dev), ldx, nullptr, 0, gamma_dev, beta_dev, eps, (half_t*)y_dev, ldy, y32_dev, ldy32, rows, D,
                                    (hipStream_t)stream);
    if (e != hipSuccess) return cgpt_fail(e == hipErrorInvalidValue ? CGPT_ERR_INVALID : CGPT_ERR_HIP,
                                          std::string("cgpt_layernorm: ") + hipGetErrorString(e));
    return CGPT_OK;
}

}  // extern "C"
