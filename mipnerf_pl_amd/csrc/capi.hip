// C ABI of libmipnerf_hip.so (see include/mipnerf_hip.h): context, weight packing, per-stage
// entry points and the level loop of MipNerf.forward (models/mip_nerf.py:172-248).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <algorithm>
#include <vector>

#include "../../include/mipnerf_hip.h"
#include "kernels.hpp"
#include "mlp_variants_gen.hpp"
#include "mlp_train_variants_gen.hpp"
#include "mlp_f32r_variants_gen.hpp"
#include "mlp_pre_variants_gen.hpp"
#include "mlp_plan_gen.hpp"

// binary tables of the training kernels (mlp_train_plan.TrainPlan.blob()), linked in through train_tables.c (.incbin)

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(MIPNERF_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

using mip::plan::PlanDesc;

// ---- plan expansion (mirror of mlp_plan.Plan.pack_table / bias_table / pack_table_f32) ----------
struct Tables {
    std::vector<int32_t> pack_bf16;   // [kNumChunks*512] flat parameter index or -1
    std::vector<int32_t> bias;        // [kNumTiles*32]
    std::vector<int32_t> pack_f32;    // [kNumChunks*512]
    std::vector<int> tensor_off;      // flat offset of each parameter tensor
    int total_params = 0;
    mip::F32Net net;
};

int kmap(int kind, int ksl, int hi, int j) {
    if (kind == 0) return ksl * 16 + hi * 8 + j;
    const int t = ksl >> 1, u = ksl & 1;
    return 32 * t + 8 * (2 * u + (j >> 2)) + 4 * hi + (j & 3);
}

void build_tables(Tables& T, const PlanDesc& P) {
    using namespace mip::plan;
    const OpDesc* kOps = P.ops;
    const int kNumOps = P.num_ops, kNetWidth = P.net_width, kXyzDim = P.xyz_dim;
    T.tensor_off.resize(P.num_param_tensors);
    int off = 0;
    for (int i = 0; i < P.num_param_tensors; ++i) { T.tensor_off[i] = off; off += P.param_numel[i]; }
    T.total_params = off;
    T.pack_bf16.assign((size_t)P.num_chunks * 512, -1);      // includes the zero padding chunks at the end of the stream
    T.pack_f32.assign((size_t)P.num_chunks * 512, -1);
    T.bias.assign((size_t)P.num_tiles * 32, -1);
    size_t ci = 0;
    auto fill_chunk = [&](const OpDesc& op, int ti, int ks) {
        const TileDesc& tile = op.tiles[ti];
        int ksl = ks, si = 0;
        while (ksl >= op.segs[si].nk) { ksl -= op.segs[si].nk; ++si; }
        const SegDesc& seg = op.segs[si];
        for (int hi = 0; hi < 2; ++hi)
            for (int j = 0; j < 8; ++j) {
                const int c = kmap(seg.kind, ksl, hi, j);
                if (c >= seg.ncols) continue;
                for (int m = 0; m < tile.nrows; ++m)
                    T.pack_bf16[ci * 512 + (size_t)(hi * 32 + m) * 8 + j] =
                        T.tensor_off[tile.wt] + (tile.row0 + m) * tile.ld + seg.col0 + c;
            }
        ++ci;
    };
    for (int oi = 0; oi < kNumOps; ++oi) {
        const OpDesc& op = kOps[oi];
        int nk = 0;
        for (int s = 0; s < op.nsegs; ++s) nk += op.segs[s].nk;
        for (int t = 0; t < op.ntiles; t += 2) {
            const bool pair = t + 1 < op.ntiles;
            if (kChainOrder) {
                for (int ks = 0; ks < nk; ++ks) fill_chunk(op, t, ks);
                if (pair) for (int ks = 0; ks < nk; ++ks) fill_chunk(op, t + 1, ks);
                continue;
            }
            for (int ks = 0; ks < nk; ++ks) {
                fill_chunk(op, t, ks);
                if (pair) fill_chunk(op, t + 1, ks);
            }
        }
        for (int t = 0; t < op.ntiles; ++t)
            for (int hi = 0; hi < 2; ++hi)
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row < op.tiles[t].nrows)
                        T.bias[(size_t)(op.first_tile + t) * 32 + hi * 16 + r] =
                            T.tensor_off[op.tiles[t].bt] + op.tiles[t].row0 + row;
                }
    }
    // fp32 stream: [op][tile][kb], natural column order; LDS layout of kernels_mlp_f32.hip: [buffer B | buffer A | encoding]
    mip::F32Net& net = T.net;
    memset(&net, 0, sizeof net);
    net.nlayers = kNumOps;
    net.width = kNetWidth;
    net.xyz_dim = kXyzDim;
    // Wide encodings (the 672 off-axis features of the unbounded-scene model): a 64-sample tile with the encoding resident does not
    // fit the CU's LDS.  Instead of halving the tile (half the reuse of every weight chunk), the two layers that read the encoding
    // stream that B operand from global memory (it stays in L1 / L2: 64 rows x 64 B per k block), and the encoding columns shrink to
    // the view features + the VALU heads' partials.
    const int ecols_full = kXyzDim > 32 ? kXyzDim : 32;
    const bool stream_enc = mip::mlp_f32_tile_samples(2 * kNetWidth + ecols_full + 4) < 64 && mip::mlp_f32_tile_samples(2 * kNetWidth + 64 + 4) == 64;
    const int ecols = stream_enc ? 64 : ecols_full;
    net.pad = stream_enc ? 1 : 0;
    net.enc_col = 2 * kNetWidth;
    net.dens_col = 2 * kNetWidth + ecols;
    net.num_rgb = P.num_rgb;
    net.ldx = 2 * kNetWidth + ecols + 4;
    size_t cf = 0;
    int cur_col = -1;                  // LDS column of the buffer that holds the current activation (mlp_plan.f32_layers)
    for (int oi = 0; oi < kNumOps; ++oi) {
        const OpDesc& op = kOps[oi];
        std::vector<int> colmap;
        for (int s = 0; s < op.nsegs; ++s)
            for (int c = 0; c < op.segs[s].nk * 16; ++c)
                colmap.push_back(c < op.segs[s].ncols ? op.segs[s].col0 + c : -1);
        const int kb = (int)colmap.size() / 16;
        mip::F32Layer& L = net.layers[oi];
        const int out_col = cur_col == kNetWidth ? 0 : kNetWidth;    // write the buffer that is not being read
        const int prev_out = cur_col;
        auto seg_col = [&](const SegDesc& sg) { return sg.kind == 0 ? net.enc_col : prev_out; };   // natural = encoding / view
        L.x_in0 = seg_col(op.segs[0]);
        L.kb0 = op.segs[0].nk;
        L.x_in1 = op.nsegs > 1 ? seg_col(op.segs[1]) : 0;
        L.kb1 = op.nsegs > 1 ? op.segs[1].nk : 0;
        L.x_out = out_col;
        L.ntiles = op.ntiles;
        L.first_tile = op.first_tile;
        L.relu = op.relu;
        L.kind = op.kind;
        L.chunk0 = (int)cf;
        L.stage_view = (op.kind == 1 && P.use_viewdirs) ? 1 : 0;
        // bit 0 / 1: K segment 0 / 1 is the sample encoding and is streamed from global memory (see above)
        L.pad = (stream_enc && op.segs[0].kind == 0 && op.segs[0].ncols == kXyzDim ? 1 : 0) |
                (stream_enc && op.nsegs > 1 && op.segs[1].kind == 0 && op.segs[1].ncols == kXyzDim ? 2 : 0);
        if (op.kind == 0 || (op.kind == 1 && op.ntiles > 1)) cur_col = out_col;     // a density-only head moves nothing
        for (int t = 0; t < op.ntiles; ++t)
            for (int k = 0; k < kb; ++k) {
                const TileDesc& tile = op.tiles[t];
                for (int hi = 0; hi < 2; ++hi)
                    for (int j = 0; j < 8; ++j) {
                        const int col = colmap[k * 16 + hi * 8 + j];
                        if (col < 0) continue;
                        for (int m = 0; m < tile.nrows; ++m)
                            T.pack_f32[cf * 512 + (size_t)(hi * 32 + m) * 8 + j] =
                                T.tensor_off[tile.wt] + (tile.row0 + m) * tile.ld + col;
                    }
                ++cf;
            }
    }
}

bool max_deg_span_is_16(const mipnerf_config& cfg) { return cfg.max_deg_point - cfg.min_deg_point == 16 && cfg.min_deg_point >= 0 && cfg.max_deg_point <= 31; }

int off_total(const Tables& T) { return T.total_params; }

// flat index -> (tensor << 20 | offset) as consumed by k_pack
std::vector<int32_t> encode(const std::vector<int32_t>& flat, const std::vector<int>& toff) {
    std::vector<int32_t> out(flat.size());
    for (size_t i = 0; i < flat.size(); ++i) {
        const int32_t f = flat[i];
        if (f < 0) { out[i] = -1; continue; }
        int t = (int)toff.size() - 1;
        while (toff[t] > f) --t;
        out[i] = (t << 20) | (f - toff[t]);
    }
    return out;
}

// ---- training tables: binary blob produced by mlp_train_plan.TrainPlan.blob(), linked in by train_tables.c ------
struct TrainTables {
    int n_bchunks, njobs, NH, NG, NMASK, NE, job_floats, nparams, n_scratch;     // NE: encoding blocks per wave tile (pre-GEMM plans), else 0
    int off_extra_w, off_extra_b, off_view_w, off_view_b;
    const int32_t* bpack;    // [n_bchunks * 512] flat parameter index or -1
    const int32_t* jobs;     // [njobs * 20]
    const int32_t* otab;     // [njobs * job_floats] flat parameter index or -1
};

bool train_tables(TrainTables& T, int variant = 0) {
    if (variant < 0 || variant >= mip::plan::kNumVariants || !mip::kTrainTableBlobs[variant]) return false;
    const int32_t* h = reinterpret_cast<const int32_t*>(mip::kTrainTableBlobs[variant]);
    if (h[0] != 0x54524E31) return false;
    T.n_bchunks = h[1]; T.njobs = h[2]; T.NH = h[3]; T.NG = h[4]; T.NMASK = h[5] & 0xffff; T.NE = h[5] >> 16; T.job_floats = h[6]; T.nparams = h[7];
    T.n_scratch = h[11]; T.off_extra_w = h[12]; T.off_extra_b = h[13]; T.off_view_w = h[14]; T.off_view_b = h[15];
    T.bpack = h + 16;
    T.jobs = T.bpack + h[8];
    T.otab = T.jobs + h[9];
    return h[9] == T.njobs * 20 && h[10] == T.njobs * T.job_floats && T.job_floats == mip::kWgradJobFloats;
}

// ---- register-resident fp32 kernel: blob produced by mlp_f32r_plan.F32RPlan.blob(), linked in by train_tables.c ------
struct F32RTables {
    int n_chunks = 0, n_aux = 0, n_groups = 0;
    const int32_t* pack = nullptr;   // [n_chunks * 256] flat parameter index or -1
    const int32_t* aux = nullptr;    // [n_aux]
};
bool f32r_tables(F32RTables& T, int variant, int nparams) {
    if (variant < 0 || variant >= mip::plan::kNumVariants || !mip::kF32RTableBlobs[variant]) return false;
    const int32_t* h = reinterpret_cast<const int32_t*>(mip::kF32RTableBlobs[variant]);
    if (h[0] != 0x46335231 || h[4] != h[1] * 256 || h[5] != nparams) return false;
    T.n_chunks = h[1]; T.n_aux = h[3]; T.n_groups = h[6];
    T.pack = h + 16;
    T.aux = T.pack + h[4];
    return true;
}

// ---- two-kernel bf16 form of wide encodings: blob produced by mlp_pre_plan.PrePlan.blob(), linked in by train_tables.c ------
struct PreTables {
    int n_gemm_chunks = 0, n_gemm_bias = 0, n_trunk_chunks = 0, n_trunk_tiles = 0, nk = 0;
    const int32_t* gemm_pack = nullptr;    // [n_gemm_chunks * 512] flat parameter index or -1
    const int32_t* gemm_bias = nullptr;    // [n_gemm_bias]
    const int32_t* trunk_pack = nullptr;   // [n_trunk_chunks * 512]
    const int32_t* trunk_bias = nullptr;   // [n_trunk_tiles * 32]
    // round 6: the one-kernel form of the same model (mlp_plan.Plan.build(arch, fused=True)), tables behind the two-kernel form's
    int n_fused_chunks = 0, n_fused_tiles = 0;
    const int32_t* fused_pack = nullptr;   // [n_fused_chunks * 512]
    const int32_t* fused_bias = nullptr;   // [n_fused_tiles * 32]
};
bool pre_tables(PreTables& T, int variant, int nparams) {
    if (variant < 0 || variant >= mip::plan::kNumVariants || !mip::kPreTableBlobs[variant]) return false;
    const int32_t* h = reinterpret_cast<const int32_t*>(mip::kPreTableBlobs[variant]);
    if (h[0] != 0x50524731 || h[7] != nparams || h[1] != h[2]) return false;
    T.n_gemm_chunks = h[1]; T.n_gemm_bias = h[3]; T.n_trunk_chunks = h[4]; T.n_trunk_tiles = h[6]; T.nk = h[8];
    T.gemm_pack = h + 16;
    T.gemm_bias = T.gemm_pack + (size_t)h[1] * 512;
    T.trunk_pack = T.gemm_bias + h[3];
    T.trunk_bias = T.trunk_pack + (size_t)h[4] * 512;
    T.n_fused_chunks = h[10]; T.n_fused_tiles = h[11];
    T.fused_pack = T.trunk_bias + (size_t)h[6] * 32;
    T.fused_bias = T.fused_pack + (size_t)h[10] * 512;
    return true;
}

}  // namespace

struct mipnerf_ctx {
    mipnerf_config cfg;
    const PlanDesc* P = nullptr;     // the generated architecture variant this context runs (mlp_plan_gen.hpp kPlans)
    Tables tab;
    int32_t* d_pack_bf16 = nullptr;
    int32_t* d_pack_f32 = nullptr;
    int32_t* d_bias_idx = nullptr;
    void* d_stream_bf16 = nullptr;   // kNumChunks * 1 KiB
    float* d_stream_f32 = nullptr;   // kNumChunks * 2 KiB
    float* d_bias = nullptr;         // kNumTiles * 32 floats
    // register-resident fp32 kernel (variants up to 256 wide): its own weight stream (1-KiB chunks) and aux table
    F32RTables f32r;
    int32_t* d_pack_f32r = nullptr;
    int32_t* d_aux_idx_f32r = nullptr;
    float* d_stream_f32r = nullptr;
    float* d_aux_f32r = nullptr;
    int f32_resident = 1;            // option 5: 1 = k_mlp_f32r where generated (inference), 0 = the LDS-resident k_mlp_f32
    // two-kernel bf16 inference of the variants whose encoding does not fit k_mlp_bf16's wave-private LDS area (gen_pre_gemm.py):
    // k_pre_gemm's weight stream + accumulator images, the trunk kernel's stream + bias table, and scratch for the per-stage entry point
    PreTables pre;
    int32_t* d_pre_idx[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};     // index tables: gemm pack, gemm bias, trunk pack, trunk bias, one-kernel pack, one-kernel bias
    void* d_fused_stream = nullptr;  // one-kernel form (round 6): its weight stream and bias table
    float* d_fused_bias = nullptr;
    int fused_pre = 1;               // option 6: 1 = mipnerf_forward runs the one-kernel form where generated, 0 = k_pre_gemm + trunk (same bits)
    void* d_pre_gemm_stream = nullptr;
    float* d_pre_gemm_bias = nullptr;
    void* d_pre_trunk_stream = nullptr;
    float* d_pre_trunk_bias = nullptr;
    void* d_pre_scratch = nullptr;   // pre_x | pre_acc of mipnerf_mlp_forward (mipnerf_forward carves them out of the caller's workspace)
    size_t pre_scratch_bytes = 0;
    hipEvent_t pre_scratch_event = nullptr;     // recorded behind the last user's kernels: another stream waits for it before it overwrites the buffer
    void* pre_scratch_stream = nullptr;
    bool pre_scratch_used = false;
    // training (bf16): W^T stream of the dgrad kernel, wgrad job tables
    TrainTables tt;
    int32_t* d_pack_dgrad = nullptr;
    int32_t* d_pack_extraT = nullptr;   // index table of W_extra^T (the transpose as a gather, so it joins the one pack launch)
    void* d_stream_dgrad = nullptr;
    mip::WgradJob* d_jobs = nullptr;
    int32_t* d_otab = nullptr;
    int4* d_wgtab = nullptr;
    int2* d_jobslots = nullptr;
    float* d_scratch = nullptr;      // fp32 scratch behind the parameters (M, db_view of one backward call)
    float* d_extra_wT = nullptr;     // W_extra^T (fp32), refreshed by mipnerf_set_params
    int num_wgrad_wgs = 0;
    mip::ParamPtrs pp;               // device pointers of the fp32 master parameters (last mipnerf_set_params)
    bool params_set = false;
    int mlp_dma = 1;                 // 1: global_load_lds ring, 0: register-staged ring (debug)
    int fused_ipe = 1;               // bf16 mipnerf_forward: IPE computed inside the MLP kernel (0: k_cast_ipe + enc buffer)
    int fuse_small = 1;              // mipnerf_forward: k_ray_prologue / k_composite_resample instead of one launch per stage (option 4)
    int grid_limit = 256;            // persistent workgroups of the bf16 MLP kernel (= CUs)
    // optional instrumentation: HIP events around every MLP launch made by mipnerf_forward
    int time_mlp = 0;
    std::vector<hipEvent_t> ev;      // pairs (start, stop)
    size_t ev_used = 0;
};

namespace {
// the bf16 inference kernel generated for this context's architecture variant
hipError_t launch_bf16_variant(mipnerf_ctx* c, const void* enc, const void* viewenc, float* rgb_sigma, float* raw, int64_t M, int N,
                               bool dma, const mip::RayInputs* rays, const float* dnoise, hipStream_t st) {
    // dnoise: density-noise draws of the level being evaluated, an ARGUMENT (not context state): two host threads / streams
    // driving the same context cannot see each other's pointer
    const mip::LaunchBf16Fn fn = mip::kLaunchBf16[c->P->variant];
    if (!fn) return hipErrorInvalidValue;          // fp32-only architecture variant (callers check has_bf16 first)
    return fn(c->d_stream_bf16, c->d_bias, enc, viewenc, rgb_sigma, raw, M, N, c->cfg.density_bias, c->cfg.rgb_padding,
              c->grid_limit, dma, rays, dnoise, c->cfg.density_noise, st);
}

// ... and its training kernels (variants whose row of the generated kLaunchTrainFwd table is not null)
static inline bool has_bf16_train(const PlanDesc* P) { return mip::kLaunchTrainFwd[P->variant] != nullptr; }
// round 5: the training form of the two-kernel bf16 MLP (wide encodings: the unbounded-scene model) -- k_pre_gemm + a trunk forward-with-save,
// the standard dgrad, weight-gradient jobs that read the row-major encoding (mlp_train_plan.TrainPlan.build(arch, pre_gemm=True))
static inline bool has_bf16_train_pre(const PlanDesc* P) { return mip::kLaunchTrainFwdPre[P->variant] != nullptr; }
static inline bool has_bf16_train_any(const PlanDesc* P) { return has_bf16_train(P) || has_bf16_train_pre(P); }
static inline bool has_bf16_pre(const PlanDesc* P) { return mip::kLaunchPreGemm[P->variant] != nullptr; }      // the two-kernel form
static inline bool has_bf16(const PlanDesc* P) { return mip::kLaunchBf16[P->variant] != nullptr || has_bf16_pre(P); }
// bytes of k_pre_gemm's two outputs for M samples: 16 KiB (X fragments) + 32 KiB (accumulator images) per wave tile, whole 256-sample tiles
static inline size_t pre_x_bytes(int64_t M) { return (size_t)((M + 255) / 256) * 8 * 16384; }
static inline size_t pre_acc_bytes(int64_t M) { return (size_t)((M + 255) / 256) * 8 * 32768; }
// bf16 B-operand fragments of the encoding: whole 256-sample tiles
static size_t pre_frag_bytes(const mipnerf_ctx* c, size_t M) { return ((M + 255) / 256) * 256 * (size_t)c->P->xyz_dim * 2; }
// k_pre_gemm + the trunk kernel.  enc: bf16, row-major [M, xyz_dim] (frag = 0) or the fragment layout launch_cast_ipe_360 writes
hipError_t launch_bf16_pre(mipnerf_ctx* c, const void* enc, int frag, const void* viewenc, float* rgb_sigma, float* raw, int64_t M, int N,
                           void* pre_x, void* pre_acc, const float* dnoise, hipStream_t st) {
    // round 6: fragment encodings go through ONE kernel where it was generated (layer 0 and the skip layer as k-step-major ops of the trunk
    // kernel, the encoding streamed through a wave-private LDS ring): no pre_x / pre_acc hand-off through HBM, same bits (option 6 = 0: two kernels)
    if (frag && c->fused_pre && c->d_fused_stream && mip::kLaunchBf16Fused[c->P->variant])
        return mip::kLaunchBf16Fused[c->P->variant](c->d_fused_stream, c->d_fused_bias, enc, nullptr, viewenc, rgb_sigma, raw, M, N,
                                                    c->cfg.density_bias, c->cfg.rgb_padding, c->grid_limit, dnoise, c->cfg.density_noise, st);
    hipError_t er = mip::kLaunchPreGemm[c->P->variant](c->d_pre_gemm_stream, c->d_pre_gemm_bias, enc, frag, pre_x, pre_acc, M, c->grid_limit, st);
    if (er != hipSuccess) return er;
    return mip::kLaunchBf16Pre[c->P->variant](c->d_pre_trunk_stream, c->d_pre_trunk_bias, pre_x, pre_acc, viewenc, rgb_sigma, raw, M, N,
                                              c->cfg.density_bias, c->cfg.rgb_padding, c->grid_limit, dnoise, c->cfg.density_noise, st);
}
hipError_t launch_trainfwd_variant(mipnerf_ctx* c, const void* enc, const void* viewenc, float* rgb_sigma, float* raw, void* act,
                                   void* masks, int64_t M, int N, const mip::RayInputs* rays, const float* dnoise, hipStream_t st) {
    const mip::LaunchTrainFwdFn fn = mip::kLaunchTrainFwd[c->P->variant];
    if (!fn) return hipErrorInvalidValue;
    return fn(c->d_stream_bf16, c->d_bias, enc, viewenc, rgb_sigma, raw, act, masks, M, N, c->cfg.density_bias, c->cfg.rgb_padding,
              c->grid_limit, rays, dnoise, c->cfg.density_noise, st);
}
hipError_t launch_dgrad_variant(mipnerf_ctx* c, const float* d_raw, const void* masks, void* delta, int64_t M, hipStream_t st) {
    const mip::LaunchDgradFn fn = mip::kLaunchDgrad[c->P->variant];
    if (!fn) return hipErrorInvalidValue;
    return fn(c->d_stream_dgrad, d_raw, masks, delta, M, c->grid_limit, st);
}

// split-K factor of the sample-contracted fp32 GEMMs: 1 x 2(3) output tiles (256 x 128) x 256 splits = 512+ eight-wave workgroups (with
// 64 splits every CU ran ONE 4-wave workgroup and nothing covered its barriers: 1.24 ms per 256 x 256 x 524288 wgrad)
constexpr int kF32WgradSplits = 256;

// the fp32 kernel evaluates the two thin heads (density, colour) on the VALU straight from the fp32 master parameters
mip::F32Net f32net_with_heads(const mipnerf_ctx* c) {
    mip::F32Net net = c->tab.net;
    const int D = c->P->net_depth, Dc = c->P->net_depth_cond;
    net.dens_w = c->pp.p[2 * D];
    net.dens_b = c->pp.p[2 * D + 1];
    net.col_w = c->pp.p[2 * D + 4 + 2 * Dc];
    net.col_b = c->pp.p[2 * D + 5 + 2 * Dc];
    return net;
}

#define NEED_BF16_TRAIN(what)                                                                                               \
    if (!has_bf16_train_any(c->P))                                                                                           \
        return fail(MIPNERF_E_UNSUPPORTED, what ": no bf16 training kernels were generated for this architecture variant " \
                                                "(gen_mlp_train.train_variants); train this shape in fp32 precision")

struct TrainWs {                   // carve-up of the caller's workspace (all 256-byte aligned)
    char* base;
    size_t off = 0;
    template <typename T> T* take(size_t bytes) {
        T* p = reinterpret_cast<T*>(base + off);
        off += align256(bytes);
        return p;
    }
};
}  // namespace

extern "C" {

int mipnerf_set_wgrad_splits(mipnerf_ctx* c, const int32_t* splits_host);

const char* mipnerf_last_error(void) { return g_err.c_str(); }
int mipnerf_abi_version(void) { return MIPNERF_ABI_VERSION; }

static void variant_to_cfg(const PlanDesc& P, mipnerf_config* cfg) {
    memset(cfg, 0, sizeof *cfg);
    cfg->num_samples = 128; cfg->num_levels = 2; cfg->min_deg_point = 0; cfg->max_deg_point = P.xyz_dim / P.feat_per_deg;
    cfg->unbounded = P.feat_per_deg == 42;
    cfg->deg_view = (P.view_dim - 3) / 6; cfg->use_viewdirs = P.use_viewdirs; cfg->net_depth = P.net_depth; cfg->net_width = P.net_width;
    cfg->net_depth_condition = P.net_depth_cond; cfg->net_width_condition = P.net_width_cond; cfg->skip_index = P.skip_index;
    cfg->num_rgb_channels = P.num_rgb; cfg->num_density_channels = P.num_density;
    cfg->resample_padding = 0.01f; cfg->density_bias = -1.0f; cfg->rgb_padding = 0.001f; cfg->density_noise = 0.0f;
}

int mipnerf_compiled_arch(mipnerf_config* cfg) {
    if (!cfg) return fail(MIPNERF_E_INVALID, "cfg is null");
    variant_to_cfg(mip::plan::kPlans[0], cfg);
    return MIPNERF_OK;
}

int mipnerf_num_variants(void) { return mip::plan::kNumVariants; }

int mipnerf_variant_arch(int variant, mipnerf_config* cfg, int* has_bf16_training) {
    if (!cfg || variant < 0 || variant >= mip::plan::kNumVariants) return fail(MIPNERF_E_INVALID, "variant_arch: bad argument");
    variant_to_cfg(mip::plan::kPlans[variant], cfg);
    if (has_bf16_training) *has_bf16_training = has_bf16_train_any(&mip::plan::kPlans[variant]);
    return MIPNERF_OK;
}

int mipnerf_create(const mipnerf_config* cfg, mipnerf_ctx** out) {
    using namespace mip::plan;
    if (!cfg || !out) return fail(MIPNERF_E_INVALID, "null argument");
#ifdef MIPNERF_EXPERIMENT_BUILD
    // build.py compiled this library with timing-experiment knobs that produce WRONG results (ablated barriers / operand reads /
    // transposing MFMAs ...): it refuses to serve unless the process says it knows
    if (!getenv("MIPNERF_ALLOW_EXPERIMENT_LIB"))
        return fail(MIPNERF_E_UNSUPPORTED, "this library is a timing-experiment build (" MIPNERF_EXPERIMENT_BUILD "): its results are wrong by "
                                           "construction; set MIPNERF_ALLOW_EXPERIMENT_LIB=1 to time it, or rebuild without those variables");
#endif
    if (cfg->num_samples < 1 || cfg->num_samples > MIPNERF_MAX_SAMPLES)
        return fail(MIPNERF_E_INVALID, "num_samples must be in [1, %d]", MIPNERF_MAX_SAMPLES);
    if (cfg->num_levels < 1 || cfg->num_levels > 2) return fail(MIPNERF_E_UNSUPPORTED, "num_levels must be 1 or 2");
    if (!cfg->use_viewdirs && cfg->net_width_condition != cfg->net_width)
        return fail(MIPNERF_E_UNSUPPORTED, "use_viewdirs=False feeds the trunk output (net_width=%d) to color_layer, which has "
                                           "net_width_condition=%d inputs: the reference MLP fails on this shape too "
                                           "(models/mip_nerf.py:99-110)", cfg->net_width, cfg->net_width_condition);
    const PlanDesc* P = nullptr;
    for (int v = 0; v < kNumVariants && !P; ++v) {
        const PlanDesc& q = kPlans[v];
        if (cfg->net_depth == q.net_depth && cfg->net_width == q.net_width && cfg->net_depth_condition == q.net_depth_cond &&
            cfg->net_width_condition == q.net_width_cond && cfg->skip_index == q.skip_index && cfg->num_rgb_channels == q.num_rgb &&
            cfg->num_density_channels == q.num_density && (cfg->unbounded ? 42 : 6) == q.feat_per_deg &&
            q.feat_per_deg * (cfg->max_deg_point - cfg->min_deg_point) == q.xyz_dim &&
            3 + 6 * cfg->deg_view == q.view_dim && (cfg->use_viewdirs != 0) == (q.use_viewdirs != 0))
            P = &q;
    }
    if (!P) {
        std::string have;
        for (int v = 0; v < kNumVariants; ++v) {
            char b[160];
            snprintf(b, sizeof b, "%s[depth %d width %d cond %dx%d skip %d xyz %d (%d per degree) view %d viewdirs %d%s]", v ? ", " : "", kPlans[v].net_depth,
                     kPlans[v].net_width, kPlans[v].net_depth_cond, kPlans[v].net_width_cond, kPlans[v].skip_index, kPlans[v].xyz_dim,
                     kPlans[v].feat_per_deg, kPlans[v].view_dim, kPlans[v].use_viewdirs, has_bf16(&kPlans[v]) ? "" : ", fp32 only");
            have += b;
        }
        return fail(MIPNERF_E_UNSUPPORTED, "no kernels / tables were generated for this MLP shape; generated: %s.  Add the shape to "
                                           "VARIANTS in csrc/gen_mlp_bf16.py and rebuild", have.c_str());
    }
    if (P->num_param_tensors > mip::kMaxParamTensors)
        return fail(MIPNERF_E_UNSUPPORTED, "variant has %d parameter tensors, the library handles up to %d", P->num_param_tensors, mip::kMaxParamTensors);
    mipnerf_ctx* c = new mipnerf_ctx();
    c->cfg = *cfg;
    c->P = P;
    build_tables(c->tab, *P);
    const std::vector<int32_t> e_bf16 = encode(c->tab.pack_bf16, c->tab.tensor_off);
    const std::vector<int32_t> e_f32 = encode(c->tab.pack_f32, c->tab.tensor_off);
    const std::vector<int32_t> e_bias = encode(c->tab.bias, c->tab.tensor_off);
    const size_t nst = (size_t)P->num_chunks * 512;
    hipError_t er = hipSuccess;
    auto chk = [&](hipError_t e) { if (er == hipSuccess) er = e; };
    chk(hipMalloc(&c->d_pack_bf16, nst * 4));
    chk(hipMalloc(&c->d_pack_f32, nst * 4));
    chk(hipMalloc(&c->d_bias_idx, e_bias.size() * 4));
    chk(hipMalloc(&c->d_stream_bf16, nst * 2));
    chk(hipMalloc(&c->d_stream_f32, nst * 4));
    chk(hipMalloc(&c->d_bias, e_bias.size() * 4));
    if (er == hipSuccess) {
        chk(hipMemcpy(c->d_pack_bf16, e_bf16.data(), nst * 4, hipMemcpyHostToDevice));
        chk(hipMemcpy(c->d_pack_f32, e_f32.data(), nst * 4, hipMemcpyHostToDevice));
        chk(hipMemcpy(c->d_bias_idx, e_bias.data(), e_bias.size() * 4, hipMemcpyHostToDevice));
    }
    if (er != hipSuccess) {
        mipnerf_destroy(c);
        return fail(MIPNERF_E_HIP, "mipnerf_create: %s", hipGetErrorString(er));
    }
    if (mip::kLaunchF32R[P->variant]) {
        if (!f32r_tables(c->f32r, P->variant, off_total(c->tab))) {
            mipnerf_destroy(c);
            return fail(MIPNERF_E_INVALID, "mipnerf_create: embedded tables of the register-resident fp32 kernel are inconsistent with the compiled plan");
        }
        const size_t np = (size_t)c->f32r.n_chunks * 256, na = (size_t)c->f32r.n_aux;
        const std::vector<int32_t> e_p = encode(std::vector<int32_t>(c->f32r.pack, c->f32r.pack + np), c->tab.tensor_off);
        const std::vector<int32_t> e_a = encode(std::vector<int32_t>(c->f32r.aux, c->f32r.aux + na), c->tab.tensor_off);
        chk(hipMalloc(&c->d_pack_f32r, np * 4));
        chk(hipMalloc(&c->d_aux_idx_f32r, na * 4));
        chk(hipMalloc(&c->d_stream_f32r, np * 4));
        chk(hipMalloc(&c->d_aux_f32r, na * 4));
        if (er == hipSuccess) {
            chk(hipMemcpy(c->d_pack_f32r, e_p.data(), np * 4, hipMemcpyHostToDevice));
            chk(hipMemcpy(c->d_aux_idx_f32r, e_a.data(), na * 4, hipMemcpyHostToDevice));
        }
        if (er != hipSuccess) {
            mipnerf_destroy(c);
            return fail(MIPNERF_E_HIP, "mipnerf_create (fp32 register-resident tables): %s", hipGetErrorString(er));
        }
    }
    if (has_bf16_pre(P)) {
        if (!pre_tables(c->pre, P->variant, off_total(c->tab))) {
            mipnerf_destroy(c);
            return fail(MIPNERF_E_INVALID, "mipnerf_create: embedded tables of the two-kernel bf16 form are inconsistent with the compiled plan");
        }
        const PreTables& pt = c->pre;
        const int32_t* src[6] = {pt.gemm_pack, pt.gemm_bias, pt.trunk_pack, pt.trunk_bias, pt.fused_pack, pt.fused_bias};
        const size_t cnt[6] = {(size_t)pt.n_gemm_chunks * 512, (size_t)pt.n_gemm_bias, (size_t)pt.n_trunk_chunks * 512, (size_t)pt.n_trunk_tiles * 32,
                               (size_t)pt.n_fused_chunks * 512, (size_t)pt.n_fused_tiles * 32};
        for (int i = 0; i < 6; ++i) {
            if (cnt[i] == 0) continue;
            const std::vector<int32_t> enc_i = encode(std::vector<int32_t>(src[i], src[i] + cnt[i]), c->tab.tensor_off);
            chk(hipMalloc(&c->d_pre_idx[i], cnt[i] * 4));
            if (er == hipSuccess) chk(hipMemcpy(c->d_pre_idx[i], enc_i.data(), cnt[i] * 4, hipMemcpyHostToDevice));
        }
        chk(hipMalloc(&c->d_pre_gemm_stream, cnt[0] * 2));
        chk(hipMalloc(&c->d_pre_gemm_bias, cnt[1] * 4));
        chk(hipMalloc(&c->d_pre_trunk_stream, cnt[2] * 2));
        chk(hipMalloc(&c->d_pre_trunk_bias, cnt[3] * 4));
        if (cnt[4]) {
            chk(hipMalloc(&c->d_fused_stream, cnt[4] * 2));
            chk(hipMalloc(&c->d_fused_bias, cnt[5] * 4));
        }
        if (er != hipSuccess) {
            mipnerf_destroy(c);
            return fail(MIPNERF_E_HIP, "mipnerf_create (tables of the two-kernel bf16 form): %s", hipGetErrorString(er));
        }
    }
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
        c->grid_limit = cus;
    // ---- training tables (one blob per variant with generated bf16 training kernels) ----
    if (has_bf16_train_any(P) && (!train_tables(c->tt, P->variant) || c->tt.nparams != off_total(c->tab))) {
        mipnerf_destroy(c);
        return fail(MIPNERF_E_INVALID, "mipnerf_create: embedded training tables are inconsistent with the compiled plan");
    }
    if (has_bf16_train_any(P)) {
        const TrainTables& tt = c->tt;
        const std::vector<int32_t> flat(tt.bpack, tt.bpack + (size_t)tt.n_bchunks * 512);
        const std::vector<int32_t> e_dg = encode(flat, c->tab.tensor_off);
        chk(hipMalloc(&c->d_pack_dgrad, e_dg.size() * 4));
        chk(hipMalloc(&c->d_stream_dgrad, e_dg.size() * 2));
        chk(hipMalloc(&c->d_jobs, (size_t)tt.njobs * sizeof(mip::WgradJob)));
        chk(hipMalloc(&c->d_otab, (size_t)tt.njobs * tt.job_floats * 4));
        chk(hipMalloc(&c->d_jobslots, (size_t)tt.njobs * sizeof(int2)));
        chk(hipMalloc(&c->d_scratch, (size_t)(tt.n_scratch > 0 ? tt.n_scratch : 1) * 4));
        chk(hipMalloc(&c->d_extra_wT, (size_t)P->net_width * P->net_width * 4));
        const int Wn = P->net_width, t_extra = 2 * P->net_depth + 2;
        std::vector<int32_t> e_xt((size_t)Wn * Wn);
        for (int i = 0; i < Wn; ++i)
            for (int j = 0; j < Wn; ++j) e_xt[(size_t)i * Wn + j] = (int32_t)((t_extra << 20) | (j * Wn + i));     // out[i][j] = W[j][i]
        chk(hipMalloc(&c->d_pack_extraT, e_xt.size() * 4));
        if (er == hipSuccess) {
            chk(hipMemcpy(c->d_pack_extraT, e_xt.data(), e_xt.size() * 4, hipMemcpyHostToDevice));
            chk(hipMemcpy(c->d_pack_dgrad, e_dg.data(), e_dg.size() * 4, hipMemcpyHostToDevice));
            chk(hipMemcpy(c->d_jobs, tt.jobs, (size_t)tt.njobs * sizeof(mip::WgradJob), hipMemcpyHostToDevice));
            chk(hipMemcpy(c->d_otab, tt.otab, (size_t)tt.njobs * tt.job_floats * 4, hipMemcpyHostToDevice));
        }
        if (er != hipSuccess) {
            mipnerf_destroy(c);
            return fail(MIPNERF_E_HIP, "mipnerf_create (training tables): %s", hipGetErrorString(er));
        }
    }
    if (has_bf16_train_any(P)) {
        const int rc = mipnerf_set_wgrad_splits(c, nullptr);
        if (rc) { mipnerf_destroy(c); return rc; }
    }
    *out = c;
    return MIPNERF_OK;
}

int mipnerf_destroy(mipnerf_ctx* c) {
    if (!c) return MIPNERF_OK;
    (void)hipFree(c->d_pack_bf16); (void)hipFree(c->d_pack_f32); (void)hipFree(c->d_bias_idx);
    (void)hipFree(c->d_stream_bf16); (void)hipFree(c->d_stream_f32); (void)hipFree(c->d_bias);
    (void)hipFree(c->d_pack_dgrad); (void)hipFree(c->d_stream_dgrad); (void)hipFree(c->d_jobs); (void)hipFree(c->d_otab);
    (void)hipFree(c->d_wgtab); (void)hipFree(c->d_jobslots); (void)hipFree(c->d_scratch); (void)hipFree(c->d_extra_wT);
    (void)hipFree(c->d_pack_extraT);
    (void)hipFree(c->d_pack_f32r); (void)hipFree(c->d_aux_idx_f32r); (void)hipFree(c->d_stream_f32r); (void)hipFree(c->d_aux_f32r);
    for (int i = 0; i < 6; ++i) (void)hipFree(c->d_pre_idx[i]);
    (void)hipFree(c->d_fused_stream); (void)hipFree(c->d_fused_bias);
    (void)hipFree(c->d_pre_gemm_stream); (void)hipFree(c->d_pre_gemm_bias); (void)hipFree(c->d_pre_trunk_stream); (void)hipFree(c->d_pre_trunk_bias);
    (void)hipFree(c->d_pre_scratch);
    if (c->pre_scratch_event) (void)hipEventDestroy(c->pre_scratch_event);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    delete c;
    return MIPNERF_OK;
}

int mipnerf_set_option(mipnerf_ctx* c, int option, int value) {
    if (!c) return fail(MIPNERF_E_INVALID, "ctx is null");
    switch (option) {
        case 0: c->mlp_dma = value ? 1 : 0; return MIPNERF_OK;
        case 1: if (value < 1) return fail(MIPNERF_E_INVALID, "grid_limit < 1"); c->grid_limit = value; return MIPNERF_OK;
        case 2: c->time_mlp = value < 0 ? 0 : (value > 2 ? 2 : value); c->ev_used = 0; return MIPNERF_OK;
        case 3: c->fused_ipe = value ? 1 : 0; return MIPNERF_OK;
        case 4: c->fuse_small = value ? 1 : 0; return MIPNERF_OK;
        case 5: c->f32_resident = value ? 1 : 0; return MIPNERF_OK;
        case 6: c->fused_pre = value ? 1 : 0; return MIPNERF_OK;
        default: return fail(MIPNERF_E_INVALID, "unknown option %d", option);
    }
}

int mipnerf_num_param_tensors(const mipnerf_ctx* c) { return c ? c->P->num_param_tensors : -1; }

int mipnerf_set_params(mipnerf_ctx* c, const float* const* params_host, void* stream) {
    if (!c || !params_host) return fail(MIPNERF_E_INVALID, "null argument");
    const PlanDesc& P = *c->P;
    mip::ParamPtrs pp;
    memset(&pp, 0, sizeof pp);
    for (int i = 0; i < P.num_param_tensors; ++i) {
        if (!params_host[i]) return fail(MIPNERF_E_INVALID, "parameter tensor %d is null", i);
        pp.p[i] = params_host[i];
    }
    const int64_t nst = (int64_t)P.num_chunks * 512;
    // every stream of the context in ONE launch: bf16 stream, fp32 stream, bias table, and for the trainable variants the
    // transposed (dgrad) stream and W_extra^T (a gather through an index table like the others)
    mip::PackSegments sg;
    memset(&sg, 0, sizeof sg);
    bool sg_overflow = false;
    auto add = [&](const int32_t* table, int64_t n, void* out, bool bf16) {
        if (sg.n >= mip::kMaxPackSegments) { sg_overflow = true; return; }       // checked BEFORE the arrays are written
        sg.table[sg.n] = table; sg.out[sg.n] = out; sg.bf16[sg.n] = bf16 ? 1 : 0;
        sg.start[sg.n + 1] = sg.start[sg.n] + n;
        ++sg.n;
    };
    add(c->d_pack_bf16, nst, c->d_stream_bf16, true);
    add(c->d_pack_f32, nst, c->d_stream_f32, false);
    add(c->d_bias_idx, (int64_t)P.num_tiles * 32, c->d_bias, false);
    if (has_bf16_train_any(&P)) {
        add(c->d_pack_dgrad, (int64_t)c->tt.n_bchunks * 512, c->d_stream_dgrad, true);
        add(c->d_pack_extraT, (int64_t)P.net_width * P.net_width, c->d_extra_wT, false);
    }
    if (c->d_stream_f32r) {
        add(c->d_pack_f32r, (int64_t)c->f32r.n_chunks * 256, c->d_stream_f32r, false);
        add(c->d_aux_idx_f32r, (int64_t)c->f32r.n_aux, c->d_aux_f32r, false);
    }
    if (c->d_pre_gemm_stream) {
        add(c->d_pre_idx[0], (int64_t)c->pre.n_gemm_chunks * 512, c->d_pre_gemm_stream, true);
        add(c->d_pre_idx[1], (int64_t)c->pre.n_gemm_bias, c->d_pre_gemm_bias, false);
        add(c->d_pre_idx[2], (int64_t)c->pre.n_trunk_chunks * 512, c->d_pre_trunk_stream, true);
        add(c->d_pre_idx[3], (int64_t)c->pre.n_trunk_tiles * 32, c->d_pre_trunk_bias, false);
        if (c->d_fused_stream) {
            add(c->d_pre_idx[4], (int64_t)c->pre.n_fused_chunks * 512, c->d_fused_stream, true);
            add(c->d_pre_idx[5], (int64_t)c->pre.n_fused_tiles * 32, c->d_fused_bias, false);
        }
    }
    if (sg_overflow) return fail(MIPNERF_E_INVALID, "set_params: more than %d pack segments", mip::kMaxPackSegments);
    HIP_TRY(mip::launch_pack_multi(sg, pp, S(stream)));
    c->pp = pp;
    c->params_set = true;
    return MIPNERF_OK;
}

// ---- per-stage entry points -----------------------------------------------------------------------
int mipnerf_sample_along_rays(int64_t B, int32_t N, const float* nearp, const float* farp, const float* t_rand,
                              int32_t disparity, float* t_samples, void* stream) {
    if (B < 1 || N < 1 || !nearp || !farp || !t_samples) return fail(MIPNERF_E_INVALID, "sample_along_rays: bad argument");
    HIP_TRY(mip::launch_sample_along_rays(B, N, nearp, farp, t_rand, disparity, t_samples, S(stream)));
    return MIPNERF_OK;
}

int mipnerf_cast_rays(int64_t B, int32_t N, const float* t, const float* origins, const float* dirs, const float* radii,
                      float* means, float* covs, void* stream) {
    if (B < 1 || N < 1 || !t || !origins || !dirs || !radii) return fail(MIPNERF_E_INVALID, "cast_rays: bad argument");
    HIP_TRY(mip::launch_cast_rays(B, N, t, origins, dirs, radii, means, covs, S(stream)));
    return MIPNERF_OK;
}

int mipnerf_cast_ipe(int64_t B, int32_t N, int32_t min_deg, int32_t max_deg, int32_t disable_integration, const float* t,
                     const float* origins, const float* dirs, const float* radii, void* enc, int out_dtype, void* stream) {
    if (B < 1 || N < 1 || !t || !origins || !dirs || !radii || !enc) return fail(MIPNERF_E_INVALID, "cast_ipe: bad argument");
    if (max_deg - min_deg != 16 || min_deg < 0 || max_deg > 31)
        return fail(MIPNERF_E_UNSUPPORTED, "cast_ipe is generated for max_deg-min_deg == 16");
    HIP_TRY(mip::launch_cast_ipe(B, N, min_deg, max_deg, disable_integration, t, origins, dirs, radii, enc,
                                 out_dtype == MIPNERF_PREC_BF16, S(stream)));
    return MIPNERF_OK;
}

int mipnerf_integrated_pos_enc(int64_t M, int32_t min_deg, int32_t max_deg, const float* means, const float* covs,
                               void* enc, int out_dtype, void* stream) {
    if (M < 1 || !means || !covs || !enc) return fail(MIPNERF_E_INVALID, "integrated_pos_enc: bad argument");
    if (max_deg - min_deg != 16 || min_deg < 0 || max_deg > 31)
        return fail(MIPNERF_E_UNSUPPORTED, "integrated_pos_enc is generated for max_deg-min_deg == 16");
    HIP_TRY(mip::launch_integrated_pos_enc(M, min_deg, max_deg, means, covs, enc, out_dtype == MIPNERF_PREC_BF16, S(stream)));
    return MIPNERF_OK;
}

int mipnerf_pos_enc(int64_t B, int32_t deg, const float* viewdirs, void* out, int32_t ld, int out_dtype, void* stream) {
    if (B < 1 || deg < 0 || !viewdirs || !out || ld < 3 + 6 * deg) return fail(MIPNERF_E_INVALID, "pos_enc: bad argument");
    HIP_TRY(mip::launch_pos_enc(B, deg, viewdirs, out, ld, out_dtype == MIPNERF_PREC_BF16, S(stream)));
    return MIPNERF_OK;
}

static int mlp_forward_noise(mipnerf_ctx* c, int64_t M, int32_t N, const void* enc, const void* viewenc, int precision,
                             float* rgb_sigma, float* raw, const float* dnoise, void* stream);

int mipnerf_mlp_forward(mipnerf_ctx* c, int64_t M, int32_t N, const void* enc, const void* viewenc, int precision,
                        float* rgb_sigma, float* raw, void* stream) {
    return mlp_forward_noise(c, M, N, enc, viewenc, precision, rgb_sigma, raw, nullptr, stream);
}

static int mlp_forward_noise(mipnerf_ctx* c, int64_t M, int32_t N, const void* enc, const void* viewenc, int precision,
                             float* rgb_sigma, float* raw, const float* dnoise, void* stream) {
    if (!c || M < 1 || N < 1 || !enc || !viewenc || !rgb_sigma) return fail(MIPNERF_E_INVALID, "mlp_forward: bad argument");
    if (!c->params_set) return fail(MIPNERF_E_INVALID, "mlp_forward: mipnerf_set_params has not been called");
    if (precision == MIPNERF_PREC_BF16) {
        if (!has_bf16(c->P)) return fail(MIPNERF_E_UNSUPPORTED, "this architecture variant (xyz_dim %d) has fp32 kernels only", c->P->xyz_dim);
        if (has_bf16_pre(c->P)) {
            // the two-kernel form needs 1.5 KiB of scratch per sample between its kernels; this per-stage entry point has no workspace
            // argument, so the context keeps a buffer that grows on demand (hipMalloc: not capturable -- mipnerf_forward uses the caller's)
            const size_t need = pre_x_bytes(M) + pre_acc_bytes(M);
            // ONE buffer per context: a call on another stream than the previous one waits for that call's kernels (event recorded behind
            // them below) before it overwrites the buffer; a regrow waits for every user
            if (!c->pre_scratch_event) { HIP_TRY(hipEventCreateWithFlags(&c->pre_scratch_event, hipEventDisableTiming)); }
            if (need > c->pre_scratch_bytes) {
                HIP_TRY(hipStreamSynchronize(S(stream)));
                if (c->pre_scratch_used) { HIP_TRY(hipEventSynchronize(c->pre_scratch_event)); }
                (void)hipFree(c->d_pre_scratch);
                c->d_pre_scratch = nullptr; c->pre_scratch_bytes = 0;
                HIP_TRY(hipMalloc(&c->d_pre_scratch, need));
                c->pre_scratch_bytes = need;
            } else if (c->pre_scratch_used && c->pre_scratch_stream != stream) {
                HIP_TRY(hipStreamWaitEvent(S(stream), c->pre_scratch_event, 0));
            }
            HIP_TRY(launch_bf16_pre(c, enc, 0, viewenc, rgb_sigma, raw, M, N, c->d_pre_scratch, (char*)c->d_pre_scratch + pre_x_bytes(M), dnoise, S(stream)));
            HIP_TRY(hipEventRecord(c->pre_scratch_event, S(stream)));
            c->pre_scratch_stream = stream; c->pre_scratch_used = true;
            return MIPNERF_OK;
        }
        HIP_TRY(launch_bf16_variant(c, enc, viewenc, rgb_sigma, raw, M, N, c->mlp_dma != 0, nullptr, dnoise, S(stream)));
    } else if (precision == MIPNERF_PREC_FP32 && c->f32_resident && c->d_stream_f32r) {
        // register-resident kernel (generated per variant, gen_mlp_f32r.py): activations in registers, weights through an LDS ring
        HIP_TRY(mip::kLaunchF32R[c->P->variant](c->d_stream_f32r, c->d_aux_f32r, (const float*)enc, (const float*)viewenc, rgb_sigma, raw, M, N,
                                                c->cfg.density_bias, c->cfg.rgb_padding, c->grid_limit, dnoise, c->cfg.density_noise, S(stream)));
    } else if (precision == MIPNERF_PREC_FP32) {
        HIP_TRY(mip::launch_mlp_f32(f32net_with_heads(c), c->d_stream_f32, c->d_bias, (const float*)enc, (const float*)viewenc,
                                    rgb_sigma, raw, M, N, c->cfg.density_bias, c->cfg.rgb_padding, nullptr, nullptr, dnoise,
                                    c->cfg.density_noise, S(stream)));
    } else {
        return fail(MIPNERF_E_INVALID, "unknown precision %d", precision);
    }
    return MIPNERF_OK;
}

int mipnerf_volumetric_rendering(int64_t B, int32_t N, const float* rgb_sigma, const float* t, const float* dirs,
                                 int32_t white_bkgd, float* comp_rgb, float* distance, float* acc, float* weights,
                                 void* stream) {
    if (B < 1 || N < 1 || N > MIPNERF_MAX_SAMPLES || !rgb_sigma || !t || !dirs || !comp_rgb || !distance || !acc || !weights)
        return fail(MIPNERF_E_INVALID, "volumetric_rendering: bad argument");
    HIP_TRY(mip::launch_volumetric_rendering(B, N, rgb_sigma, t, dirs, white_bkgd, comp_rgb, distance, acc, weights,
                                             S(stream)));
    return MIPNERF_OK;
}

int mipnerf_resample_along_rays(int64_t B, int32_t N, const float* t, const float* weights, const float* u_rand,
                                float resample_padding, float* t_new, void* stream) {
    if (B < 1 || N < 1 || N > MIPNERF_MAX_SAMPLES || !t || !weights || !t_new)
        return fail(MIPNERF_E_INVALID, "resample_along_rays: bad argument");
    HIP_TRY(mip::launch_piecewise_constant_pdf(B, N, t, weights, N + 1, u_rand, true, resample_padding, t_new, S(stream)));
    return MIPNERF_OK;
}

int mipnerf_sorted_piecewise_constant_pdf(int64_t B, int32_t nbins, const float* bins, const float* weights,
                                          int32_t ndraws, const float* u_rand, float* samples, void* stream) {
    if (B < 1 || nbins < 1 || nbins > MIPNERF_MAX_SAMPLES || ndraws < 1 || !bins || !weights || !samples)
        return fail(MIPNERF_E_INVALID, "sorted_piecewise_constant_pdf: bad argument");
    HIP_TRY(mip::launch_piecewise_constant_pdf(B, nbins, bins, weights, ndraws, u_rand, false, 0.0f, samples, S(stream)));
    return MIPNERF_OK;
}

// ---- unbounded scenes (mip-NeRF 360): what the reference's dead code at mip.py:106-124, 292-319, 424-447 aims at ----
int mipnerf_sample_along_rays_360(int64_t B, int32_t N, const float* nearp, const float* farp, const float* t_rand,
                                  float* t_inv, float* t_samples, void* stream) {
    if (B < 1 || N < 1 || !nearp || !farp || !t_inv || !t_samples) return fail(MIPNERF_E_INVALID, "sample_along_rays_360: bad argument");
    HIP_TRY(mip::launch_sample_along_rays_360(B, N, nearp, farp, t_rand, t_inv, t_samples, S(stream)));
    return MIPNERF_OK;
}

int mipnerf_cast_ipe_360(int64_t B, int32_t N, int32_t min_deg, int32_t max_deg, int32_t contracted, const float* t,
                         const float* origins, const float* dirs, const float* radii, void* enc, int out_dtype, float* means,
                         float* covs, void* stream) {
    if (B < 1 || N < 1 || !t || !origins || !dirs || !radii || (!enc && !means) || ((means == nullptr) != (covs == nullptr)))
        return fail(MIPNERF_E_INVALID, "cast_ipe_360: bad argument");
    if (min_deg < 0 || max_deg <= min_deg || max_deg > 31) return fail(MIPNERF_E_INVALID, "cast_ipe_360: need 0 <= min_deg < max_deg <= 31");
    const bool frag = out_dtype == MIPNERF_OUT_BF16_FRAGMENTS;
    if (frag && (!enc || means)) return fail(MIPNERF_E_INVALID, "cast_ipe_360: the fragment layout is an encoding-only output");
    HIP_TRY(mip::launch_cast_ipe_360(B, N, min_deg, max_deg, contracted, t, origins, dirs, radii, enc,
                                     out_dtype == MIPNERF_PREC_BF16 || frag, means, covs, S(stream), frag));
    return MIPNERF_OK;
}

int mipnerf_gauss_360(int64_t M, int32_t min_deg, int32_t max_deg, int32_t contracted, const float* means, const float* covs,
                      void* enc, int out_dtype, float* means_out, float* covs_out, void* stream) {
    if (M < 1 || !means || !covs || (!enc && !means_out)) return fail(MIPNERF_E_INVALID, "gauss_360: bad argument");
    if (enc && (min_deg < 0 || max_deg <= min_deg || max_deg > 31)) return fail(MIPNERF_E_INVALID, "gauss_360: need 0 <= min_deg < max_deg <= 31");
    HIP_TRY(mip::launch_gauss_360(M, min_deg, enc ? max_deg : min_deg + 1, contracted, means, covs, enc, out_dtype == MIPNERF_PREC_BF16,
                                  means_out, covs_out, S(stream)));
    return MIPNERF_OK;
}

// ---- device-side ray generation (datasets/datasets.py:116-168, 214-263) ---------------------------------------
int mipnerf_generate_rays(int64_t n, const float* cameras, const int32_t* cam_idx, const int32_t* pix_idx,
                          const mipnerf_rays_out* out, void* stream) {
    if (n < 1 || !cameras || !out || !out->origins || !out->directions || !out->viewdirs || !out->radii ||
        !out->lossmult || !out->near || !out->far)
        return fail(MIPNERF_E_INVALID, "generate_rays: bad argument");
    HIP_TRY(mip::launch_generate_rays(n, cameras, cam_idx, pix_idx, out->origins, out->directions, out->viewdirs,
                                      out->radii, out->lossmult, out->near, out->far, S(stream)));
    return MIPNERF_OK;
}

int mipnerf_generate_rays_f64(int64_t n, const double* cameras, const int32_t* cam_idx, const int32_t* pix_idx,
                              const mipnerf_rays_out* out, void* stream) {
    if (n < 1 || !cameras || !out || !out->origins || !out->directions || !out->viewdirs || !out->radii ||
        !out->lossmult || !out->near || !out->far)
        return fail(MIPNERF_E_INVALID, "generate_rays_f64: bad argument");
    HIP_TRY(mip::launch_generate_rays_f64(n, cameras, cam_idx, pix_idx, out->origins, out->directions, out->viewdirs,
                                          out->radii, out->lossmult, out->near, out->far, S(stream)));
    return MIPNERF_OK;
}

// ---- evaluation metrics (utils/metrics.py:191-197) ---------------------------------------------------------------
int64_t mipnerf_eval_workspace_floats(int32_t height, int32_t width) {
    return height > 0 && width > 0 ? mip::eval_errors_partial_floats(height, width) : 0;
}

int mipnerf_eval_errors(int32_t H, int32_t W, const float* pred, const float* gt, float* workspace, float* out_psnr_ssim,
                        void* stream) {
    if (H < 1 || W < 1 || !pred || !gt || !workspace || !out_psnr_ssim) return fail(MIPNERF_E_INVALID, "eval_errors: bad argument");
    HIP_TRY(mip::launch_eval_errors(H, W, pred, gt, workspace, out_psnr_ssim, S(stream)));
    return MIPNERF_OK;
}

// ---- training-side entry points ------------------------------------------------------------------
int mipnerf_activate(int64_t M, const float* raw, float rgb_padding, float density_bias, const float* density_randn,
                     float density_noise, float* rgb_sigma, void* stream) {
    if (M < 1 || !raw || !rgb_sigma) return fail(MIPNERF_E_INVALID, "activate: bad argument");
    HIP_TRY(mip::launch_activate(M, raw, rgb_padding, density_bias, density_randn, density_noise, rgb_sigma, S(stream)));
    return MIPNERF_OK;
}

int mipnerf_volumetric_rendering_bwd(int64_t B, int32_t N, const float* rgb_sigma, const float* t, const float* dirs,
                                     int32_t white_bkgd, const float* g_rgb, const float* g_dist, const float* g_acc,
                                     const float* g_w, float rgb_padding, float* d_raw, void* stream) {
    if (B < 1 || N < 1 || N > MIPNERF_MAX_SAMPLES || !rgb_sigma || !t || !dirs || !d_raw)
        return fail(MIPNERF_E_INVALID, "volumetric_rendering_bwd: bad argument");
    HIP_TRY(mip::launch_volumetric_rendering_bwd(B, N, rgb_sigma, t, dirs, white_bkgd, g_rgb, g_dist, g_acc, g_w,
                                                 rgb_padding, d_raw, S(stream)));
    return MIPNERF_OK;
}

int mipnerf_distloss(int64_t B, int32_t N, const float* weights, const float* t, float* ray_loss, const float* g_ray,
                     float* d_w, void* stream) {
    if (B < 1 || N < 1 || N > MIPNERF_MAX_SAMPLES || !weights || !t || (!ray_loss && !d_w) || ((g_ray == nullptr) != (d_w == nullptr)))
        return fail(MIPNERF_E_INVALID, "distloss: bad argument");
    HIP_TRY(mip::launch_distloss(B, N, weights, t, ray_loss, g_ray, 0.0f, d_w, S(stream)));
    return MIPNERF_OK;
}

// ---- gradient through the resampler (stop_resample_grad=False, mip.py:265-279): the pieces autograd chains ------------------
int mipnerf_volumetric_rendering_bwd_t(int64_t B, int32_t N, const float* rgb_sigma, const float* t, const float* dirs,
                                       int32_t white_bkgd, const float* g_rgb, const float* g_dist, const float* g_acc,
                                       const float* g_w, float rgb_padding, float* d_raw, float* d_t, void* stream) {
    if (B < 1 || N < 1 || N > MIPNERF_MAX_SAMPLES || !rgb_sigma || !t || !dirs || !d_raw || !d_t)
        return fail(MIPNERF_E_INVALID, "volumetric_rendering_bwd_t: bad argument");
    HIP_TRY(mip::launch_volumetric_rendering_bwd(B, N, rgb_sigma, t, dirs, white_bkgd, g_rgb, g_dist, g_acc, g_w,
                                                 rgb_padding, d_raw, S(stream), d_t));
    return MIPNERF_OK;
}

int mipnerf_distloss_bwd(int64_t B, int32_t N, const float* weights, const float* t, const float* g_ray, float* d_w, float* d_t,
                         void* stream) {
    if (B < 1 || N < 1 || N > MIPNERF_MAX_SAMPLES || !weights || !t || !g_ray || !d_w || !d_t)
        return fail(MIPNERF_E_INVALID, "distloss_bwd: bad argument");
    HIP_TRY(mip::launch_distloss(B, N, weights, t, nullptr, g_ray, 0.0f, d_w, S(stream), d_t));
    return MIPNERF_OK;
}

int mipnerf_cast_ipe_bwd(int64_t B, int32_t N, int32_t min_deg, int32_t max_deg, int32_t disable_integration, const float* t,
                         const float* origins, const float* dirs, const float* radii, const float* d_enc, float* d_t, void* stream) {
    if (B < 1 || N < 1 || !t || !origins || !dirs || !radii || !d_enc || !d_t || min_deg < 0 || max_deg > 31 || max_deg <= min_deg)
        return fail(MIPNERF_E_INVALID, "cast_ipe_bwd: bad argument");
    HIP_TRY(mip::launch_cast_ipe_bwd(B, N, min_deg, max_deg, disable_integration, t, origins, dirs, radii, d_enc, d_t, S(stream)));
    return MIPNERF_OK;
}

int mipnerf_resample_along_rays_bwd(int64_t B, int32_t N, const float* t, const float* weights, const float* u_rand,
                                    float resample_padding, const float* d_t_new, float* d_weights, void* stream) {
    if (B < 1 || N < 1 || N > MIPNERF_MAX_SAMPLES || !t || !weights || !d_t_new || !d_weights)
        return fail(MIPNERF_E_INVALID, "resample_along_rays_bwd: bad argument");
    HIP_TRY(mip::launch_resample_bwd(B, N, t, weights, u_rand, resample_padding, d_t_new, d_weights, S(stream)));
    return MIPNERF_OK;
}

// ---- native MLP training step (bf16) ------------------------------------------------------------------
int mipnerf_mlp_train_sizes(const mipnerf_ctx* c, int64_t M, size_t* act_bytes, size_t* mask_bytes, size_t* delta_bytes,
                            size_t* partial_bytes) {
    if (!c || M < 1) return fail(MIPNERF_E_INVALID, "mlp_train_sizes: bad argument");
    NEED_BF16_TRAIN("mlp_train_sizes");
    const size_t n_wt = (size_t)((M + 255) / 256) * 8;          // wave tiles (32 samples) of whole workgroup tiles
    // pre-GEMM form: behind the T-blocks a 256-byte record of the encoding the forward ran on (the weight-gradient kernel reads it: the act
    // buffer describes itself, mipnerf_mlp_wgrad needs no extra argument) and the two buffers between k_pre_gemm and the trunk kernel
    if (act_bytes) *act_bytes = n_wt * c->tt.NH * 2048 + (has_bf16_train_pre(c->P) ? 256 + pre_x_bytes(M) + pre_acc_bytes(M) : 0);
    if (mask_bytes) *mask_bytes = n_wt * c->tt.NMASK * 1024;
    if (delta_bytes) *delta_bytes = n_wt * c->tt.NG * 2048;
    if (partial_bytes) *partial_bytes = (size_t)c->num_wgrad_wgs * c->tt.job_floats * 4;
    return MIPNERF_OK;
}

// enc_frag (two-kernel form only): enc is the B-operand fragment buffer of whole 256-sample tiles (launch_cast_ipe_360(..., frag = true)) instead of
// row-major rows -- what the one-call training step writes (k_pre_gemm reads fragments 40 % faster than rows)
static int mlp_forward_train_noise(mipnerf_ctx* c, int64_t M, int32_t N, const void* enc, const void* viewenc, float* rgb_sigma,
                                   float* raw, void* act, void* masks, const float* dnoise, void* stream, bool enc_frag = false) {
    if (!c || M < 1 || N < 1 || !enc || !viewenc || !rgb_sigma || !raw || !act || !masks)
        return fail(MIPNERF_E_INVALID, "mlp_forward_train: bad argument");
    NEED_BF16_TRAIN("mlp_forward_train");
    if (!c->params_set) return fail(MIPNERF_E_INVALID, "mlp_forward_train: mipnerf_set_params has not been called");
    if (has_bf16_train_pre(c->P)) {
        // enc: bf16 row-major [M, xyz_dim] (what mipnerf_cast_ipe_360 writes); it must stay valid until the backward of this act buffer
        const size_t n_wt = (size_t)((M + 255) / 256) * 8;
        char* rec = (char*)act + n_wt * c->tt.NH * 2048;
        char* pre_x = rec + 256;
        char* pre_acc = pre_x + pre_x_bytes(M);
        HIP_TRY(mip::kLaunchPreGemm[c->P->variant](c->d_pre_gemm_stream, c->d_pre_gemm_bias, enc, enc_frag ? 1 : 0, pre_x, pre_acc, M, c->grid_limit, S(stream)));
        HIP_TRY(mip::kLaunchTrainFwdPre[c->P->variant](c->d_pre_trunk_stream, c->d_pre_trunk_bias, pre_x, pre_acc, viewenc, rgb_sigma, raw, act, masks,
                                                        M, N, c->cfg.density_bias, c->cfg.rgb_padding, c->grid_limit, dnoise,
                                                        c->cfg.density_noise, S(stream)));
        HIP_TRY(mip::launch_wgrad_record_enc(rec, enc, M, enc_frag ? 0 : c->P->xyz_dim * 2, S(stream), c->P->xyz_dim / 16));
        return MIPNERF_OK;
    }
    HIP_TRY(launch_trainfwd_variant(c, enc, viewenc, rgb_sigma, raw, act, masks, M, N, nullptr, dnoise, S(stream)));
    return MIPNERF_OK;
}

int mipnerf_mlp_forward_train(mipnerf_ctx* c, int64_t M, int32_t N, const void* enc, const void* viewenc, float* rgb_sigma,
                              float* raw, void* act, void* masks, void* stream) {
    return mlp_forward_train_noise(c, M, N, enc, viewenc, rgb_sigma, raw, act, masks, nullptr, stream, false);
}

int mipnerf_mlp_forward_train_fragments(mipnerf_ctx* c, int64_t M, int32_t N, const void* enc, const void* viewenc, float* rgb_sigma,
                                        float* raw, void* act, void* masks, void* stream) {
    if (c && !has_bf16_train_pre(c->P))
        return fail(MIPNERF_E_UNSUPPORTED, "mlp_forward_train_fragments: this context has no two-kernel (pre-GEMM) training form; its kernels read row-major encodings");
    return mlp_forward_train_noise(c, M, N, enc, viewenc, rgb_sigma, raw, act, masks, nullptr, stream, true);
}

int mipnerf_mlp_dgrad(mipnerf_ctx* c, int64_t M, const float* d_raw, const void* masks, void* delta, void* stream) {
    if (!c || M < 1 || !d_raw || !masks || !delta) return fail(MIPNERF_E_INVALID, "mlp_dgrad: bad argument");
    NEED_BF16_TRAIN("mlp_dgrad");
    if (!c->params_set) return fail(MIPNERF_E_INVALID, "mlp_dgrad: mipnerf_set_params has not been called");
    HIP_TRY(launch_dgrad_variant(c, d_raw, masks, delta, M, S(stream)));
    return MIPNERF_OK;
}

// weight gradients of `n_wt` consecutive wave tiles (32 samples each; the tiles of several levels may be concatenated)
static int wgrad_tiles(mipnerf_ctx* c, int64_t n_wt, const void* act, const void* delta, float* partials, float* grad_flat,
                       int32_t accumulate, void* stream);

int mipnerf_mlp_wgrad(mipnerf_ctx* c, int64_t M, const void* act, const void* delta, float* partials, float* grad_flat,
                      int32_t accumulate, void* stream) {
    if (!c || M < 1 || !act || !delta || !partials) return fail(MIPNERF_E_INVALID, "mlp_wgrad: bad argument");
    NEED_BF16_TRAIN("mlp_wgrad");
    return wgrad_tiles(c, ((M + 255) / 256) * 8, act, delta, partials, grad_flat, accumulate, stream);
}

static int wgrad_tiles(mipnerf_ctx* c, int64_t n_wt, const void* act, const void* delta, float* partials, float* grad_flat,
                       int32_t accumulate, void* stream) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->time_mlp == 2) {          // option 2 = 2: time the weight-gradient launches (bench.py --mode train roofline)
        if (c->ev_used + 2 > c->ev.size())
            for (int i = 0; i < 64; ++i) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); c->ev.push_back(e); }
        e0 = c->ev[c->ev_used]; e1 = c->ev[c->ev_used + 1]; c->ev_used += 2;
        HIP_TRY(hipEventRecord(e0, S(stream)));
    }
    const mip::WgradEnc* rec = has_bf16_train_pre(c->P) ? (const mip::WgradEnc*)((const char*)act + (size_t)n_wt * c->tt.NH * 2048) : nullptr;
    HIP_TRY(mip::launch_mlp_wgrad(act, delta, c->d_jobs, c->d_wgtab, c->num_wgrad_wgs, n_wt, c->tt.NH, c->tt.NG, partials,
                                  S(stream), rec));
    if (e1) HIP_TRY(hipEventRecord(e1, S(stream)));
    if (grad_flat) {
        if (!c->params_set) return fail(MIPNERF_E_INVALID, "mlp_wgrad: mipnerf_set_params has not been called");
        const PlanDesc& P = *c->P;
        mip::WgradPost post = {};
        if (P.use_viewdirs) {
            post.W = P.net_width; post.Wc = P.net_width_cond; post.ldv = P.net_width + P.view_dim;
            post.off_extra_w = c->tt.off_extra_w; post.off_extra_b = c->tt.off_extra_b;
            post.off_view_w = c->tt.off_view_w; post.off_view_b = c->tt.off_view_b;
            // state_dict order: ... density (2*D, 2*D+1), extra (2*D+2, +3), view (2*D+4, +5), colour
            post.extra_wT = c->d_extra_wT;
            post.extra_w = c->pp.p[2 * P.net_depth + 2]; post.extra_b = c->pp.p[2 * P.net_depth + 3];
            post.view_w = c->pp.p[2 * P.net_depth + 4];
        } else if (!accumulate) {
            // MLP.forward(x, None): extra_layer / view_layers are unused parameters (autograd leaves their .grad None) and no
            // partial feeds them: zero unless accumulating
            const int D = P.net_depth;
            for (int t = 2 * D + 2; t < 2 * D + 6; ++t)
                HIP_TRY(hipMemsetAsync(grad_flat + c->tab.tensor_off[t], 0, (size_t)P.param_numel[t] * 4, S(stream)));
        }
        HIP_TRY(mip::launch_wgrad_reduce(partials, c->d_otab, c->d_jobslots, c->tt.njobs, grad_flat, c->d_scratch,
                                         c->tt.nparams, post, accumulate != 0, S(stream)));
    }
    return MIPNERF_OK;
}

int mipnerf_mlp_backward(mipnerf_ctx* c, int64_t M, const float* d_raw, const void* act, const void* masks, void* delta,
                         float* partials, float* grad_flat, int32_t accumulate, void* stream) {
    if (!grad_flat) return fail(MIPNERF_E_INVALID, "mlp_backward: grad_flat is null");
    int rc = mipnerf_mlp_dgrad(c, M, d_raw, masks, delta, stream);
    if (rc) return rc;
    return mipnerf_mlp_wgrad(c, M, act, delta, partials, grad_flat, accumulate, stream);
}

int mipnerf_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                      double beta2, double eps, int32_t step, void* stream) {
    if (n < 1 || !param || !grad || !exp_avg || !exp_avg_sq || step < 1) return fail(MIPNERF_E_INVALID, "adam_step: bad argument");
    HIP_TRY(mip::launch_adam_flat(n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, S(stream)));
    return MIPNERF_OK;
}

int mipnerf_adam_step_scheduled(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                const mipnerf_lr_schedule* sc, int64_t* step_count, float* hyper_out, void* stream) {
    if (n < 1 || !param || !grad || !exp_avg || !exp_avg_sq || !sc || !step_count || !hyper_out)
        return fail(MIPNERF_E_INVALID, "adam_step_scheduled: bad argument");
    if (sc->constant_lr <= 0.0 && (sc->lr_init <= 0.0 || sc->lr_final <= 0.0 || sc->max_steps < 1))
        return fail(MIPNERF_E_INVALID, "adam_step_scheduled: lr_init, lr_final and max_steps must be positive");
    static_assert(sizeof(mipnerf_lr_schedule) == sizeof(mip::LrSchedule), "schedule structs must match");
    mip::LrSchedule k;
    memcpy(&k, sc, sizeof k);
    HIP_TRY(mip::launch_adam_scheduled(n, param, grad, exp_avg, exp_avg_sq, k, step_count, hyper_out, S(stream)));
    return MIPNERF_OK;
}

// Replace the wgrad work split: splits_host[njobs] workgroups per job (0 = skip the job: its gradients are then
// NOT produced -- timing experiments only).  NULL restores the default (workgroups per job ~ blocks per stage + 4, all CUs handed out).
int mipnerf_set_wgrad_splits(mipnerf_ctx* c, const int32_t* splits_host) {
    if (!c) return fail(MIPNERF_E_INVALID, "ctx is null");
    NEED_BF16_TRAIN("set_wgrad_splits");
    const TrainTables& tt = c->tt;
    std::vector<int> sp(tt.njobs);
    if (splits_host) {
        for (int j = 0; j < tt.njobs; ++j) sp[j] = splits_host[j] < 0 ? 0 : splits_host[j];
    } else {
        // Workgroups per job ~ (blocks the job moves per stage) + kStageFixedBlocks: a stage costs its bytes plus a fixed part (barrier +
        // DMA latency; a job alone on 16 workgroups takes 0.09 us + 0.046 us per 2-KiB block per stage, scripts/prof_train.py
        // --experiments).  Measured on MI355X at 4096 x 128 samples per level (profiles/r03aa_wgrad_splits.txt): bytes-proportional
        // (fixed part 0) 0.954 ms, equal workgroups per job (the rounds 1-2 default; fixed part -> infinity) 0.851-0.853 ms, fixed part
        // of 4 blocks 0.817 ms -- the small jobs no longer hold 21 CUs each long after they could have finished on 11.  All of the
        // grid is handed out (largest remainders first).
        constexpr double kStageFixedBlocks = 4.0;
        std::vector<double> w(tt.njobs);
        double wsum = 0.0;
        for (int j = 0; j < tt.njobs; ++j) wsum += (w[j] = tt.jobs[j * 20 + 0] + tt.jobs[j * 20 + 1] + kStageFixedBlocks);
        int total = 0;
        std::vector<std::pair<double, int>> frac(tt.njobs);
        for (int j = 0; j < tt.njobs; ++j) {
            const double x = w[j] / wsum * c->grid_limit;
            sp[j] = (int)x < 1 ? 1 : (int)x;
            frac[j] = {x - sp[j], j};
            total += sp[j];
        }
        std::sort(frac.begin(), frac.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) {
            return a.first > b.first || (a.first == b.first && a.second < b.second); });
        for (int k = 0; total < c->grid_limit && k < tt.njobs; ++k, ++total) ++sp[frac[k].second];
    }
    std::vector<int4> wgtab;
    std::vector<int2> slots(tt.njobs);
    for (int j = 0; j < tt.njobs; ++j) {
        slots[j] = make_int2((int)wgtab.size(), sp[j]);
        for (int k = 0; k < sp[j]; ++k) wgtab.push_back(make_int4(j, k, sp[j], (int)wgtab.size()));
    }
    if (wgtab.empty()) return fail(MIPNERF_E_INVALID, "set_wgrad_splits: no workgroups");
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(c->d_wgtab);
    c->d_wgtab = nullptr;
    HIP_TRY(hipMalloc(&c->d_wgtab, wgtab.size() * sizeof(int4)));
    HIP_TRY(hipMemcpy(c->d_wgtab, wgtab.data(), wgtab.size() * sizeof(int4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_jobslots, slots.data(), slots.size() * sizeof(int2), hipMemcpyHostToDevice));
    c->num_wgrad_wgs = (int)wgtab.size();
    return MIPNERF_OK;
}



// ---- parity-mode (fp32) MLP training: fused forward that saves every layer output + GEMM-based backward ------------
// `save` = one [M, width] fp32 slot per op of the plan (shipped: layers 0..7 outputs, bottleneck, view-layer output
// [M,128] in slot 9), followed by the ReLU sign bits of the same slots (1 bit per element, f32_bits_slot_words) that the
// dgrad epilogue reads instead of the fp32 values.
size_t mipnerf_mlp_train_f32_bytes(const mipnerf_ctx* c, int64_t M, size_t* save_bytes, size_t* workspace_bytes) {
    if (!c || M < 1) return 0;
    const PlanDesc& P = *c->P;
    const size_t save = (size_t)P.num_ops * M * P.net_width * 4           // one [M, width] slot per layer (op) of the plan
                        + (size_t)P.num_ops * mip::f32_bits_slot_words(M, P.net_width) * 8;   // + their sign bits
    const int splits = kF32WgradSplits;
    const size_t wmax = P.net_width > P.net_width_cond ? P.net_width : P.net_width_cond;
    size_t part = (size_t)splits * wmax * (wmax + P.xyz_dim + 32);                 // split-K partials of the big wgrad GEMMs
    const size_t thin = (size_t)((M + 2047) / 2048) * 32 * (wmax + 1);              // per-slice partials of the thin ones
    if (thin > part) part = thin;
    const size_t ws = 2 * (size_t)M * wmax * 4 + part * 4 + 1024;
    if (save_bytes) *save_bytes = save;
    if (workspace_bytes) *workspace_bytes = ws;
    return save + ws;
}

int mipnerf_mlp_forward_train_f32(mipnerf_ctx* c, int64_t M, int32_t N, const float* enc, const float* viewenc, float* rgb_sigma,
                                  float* raw, float* save, void* stream) {
    if (!c || M < 1 || N < 1 || !enc || !viewenc || !rgb_sigma || !raw || !save)
        return fail(MIPNERF_E_INVALID, "mlp_forward_train_f32: bad argument");
    if (!c->params_set) return fail(MIPNERF_E_INVALID, "mlp_forward_train_f32: mipnerf_set_params has not been called");
    HIP_TRY(mip::launch_mlp_f32(f32net_with_heads(c), c->d_stream_f32, c->d_bias, enc, viewenc, rgb_sigma, raw, M, N, c->cfg.density_bias,
                                c->cfg.rgb_padding, save,
                                reinterpret_cast<unsigned long long*>(save + (size_t)c->P->num_ops * M * c->P->net_width), nullptr, 0.0f,
                                S(stream)));
    return MIPNERF_OK;
}

static int mlp_backward_f32_impl(mipnerf_ctx* c, int64_t M, int32_t N, const float* d_raw, const float* enc, const float* viewenc,
                                 const float* save, void* workspace, float* grad_flat, int32_t accumulate, float* d_enc, void* stream);

int mipnerf_mlp_backward_f32(mipnerf_ctx* c, int64_t M, int32_t N, const float* d_raw, const float* enc, const float* viewenc,
                             const float* save, void* workspace, float* grad_flat, int32_t accumulate, void* stream) {
    return mlp_backward_f32_impl(c, M, N, d_raw, enc, viewenc, save, workspace, grad_flat, accumulate, nullptr, stream);
}

// ... plus d_enc [M, xyz_dim] = dL/d(encoding) = delta_0 W_0 + delta_skip W_skip[:, W:] (the input gradient the reference's
// autograd produces when the encoding requires grad, i.e. with stop_resample_grad=False)
int mipnerf_mlp_backward_f32_enc(mipnerf_ctx* c, int64_t M, int32_t N, const float* d_raw, const float* enc, const float* viewenc,
                                 const float* save, void* workspace, float* grad_flat, int32_t accumulate, float* d_enc,
                                 void* stream) {
    if (!d_enc) return fail(MIPNERF_E_INVALID, "mlp_backward_f32_enc: d_enc is null");
    return mlp_backward_f32_impl(c, M, N, d_raw, enc, viewenc, save, workspace, grad_flat, accumulate, d_enc, stream);
}

static int mlp_backward_f32_impl(mipnerf_ctx* c, int64_t M, int32_t N, const float* d_raw, const float* enc, const float* viewenc,
                                 const float* save, void* workspace, float* grad_flat, int32_t accumulate, float* d_enc, void* stream) {
    if (!c || M < 1 || N < 1 || !d_raw || !enc || !viewenc || !save || !workspace || !grad_flat)
        return fail(MIPNERF_E_INVALID, "mlp_backward_f32: bad argument");
    if (!c->params_set) return fail(MIPNERF_E_INVALID, "mlp_backward_f32: mipnerf_set_params has not been called");
    if (M > 0x7fffffff) return fail(MIPNERF_E_INVALID, "mlp_backward_f32: too many samples");
    const PlanDesc& PL = *c->P;
    const int W = PL.net_width, Wc = PL.net_width_cond, E = PL.xyz_dim, D = PL.net_depth, V = PL.view_dim, RGB = PL.num_rgb;
    const bool views = PL.use_viewdirs != 0;
    const int splits = kF32WgradSplits, Mi = (int)M;
    hipStream_t st = S(stream);
    const size_t wmax = W > Wc ? W : Wc;
    float* g0 = reinterpret_cast<float*>(workspace);
    float* g1 = g0 + (size_t)M * wmax;
    float* part = g1 + (size_t)M * wmax;
    const bool acc = accumulate != 0;
    // `save` slot of op L (k_mlp_f32): [M, 32 * hidden tiles] fp32 at offset L * M * W
    auto slot = [&](int L) { return save + (size_t)L * M * W; };
    const unsigned long long* bits0 = reinterpret_cast<const unsigned long long*>(save + (size_t)PL.num_ops * M * W);
    auto slot_bits = [&](int L) { return bits0 + (size_t)L * mip::f32_bits_slot_words(M, W); };
    auto P = [&](int t) { return c->pp.p[t]; };                       // fp32 master parameter t (state_dict order)
    auto G = [&](int t) { return grad_flat + c->tab.tensor_off[t]; }; // its gradient
    const int Dc = PL.net_depth_cond;        // view layers (mip_nerf.py:62-69): the first reads [bottleneck | view encoding], the others Wc -> Wc
    const int tDensW = 2 * D, tDensB = 2 * D + 1, tExW = 2 * D + 2, tExB = 2 * D + 3, tVW = 2 * D + 4, tVB = 2 * D + 5,
              tCW = 2 * D + 4 + 2 * Dc, tCB = 2 * D + 5 + 2 * Dc;
    const float* x8 = slot(D - 1);     // trunk output [M, W]
    // wgrad / bias helpers: dW[out, ldw] (cols [col0, col0+n)) (+)= dY[M, out]^T X[M, n];  db[out] (+)= dY^T 1.
    // Shapes with >= 64 rows / columns go to the 128 x 128-tile kernel (bias gradient fused into its A-tile staging, ReLU mask
    // fused into the dgrad epilogue); thin ones (colour / density heads, the 27 view features) to the 64 x 64 kernel.
    auto wgrad = [&](const float* dY, int64_t ldy, int nout, const float* X, int64_t ldx, int rowdiv, int ncols, float* dW, int64_t ldw,
                     float* db) -> hipError_t {
        if (rowdiv == 1 && mip::gemm_f32_big_ok(nout, ncols, M, dY, ldy))
            return mip::launch_gemm_f32_big(true, nout, ncols, M, dY, ldy, X, ldx, dW, ldw, acc, splits, part, nullptr, nullptr, 0, nullptr, db, st);
        if (nout <= 4)            // colour / density heads: few output ROWS, wide in the input features
            return mip::launch_thin_wgrad(M, ncols, nout, X, ldx, dY, ldy, 1, dW, 1, ldw, db, acc, part, st);
        if (ncols <= 32 && !db)   // the view features: few input COLUMNS (one row of X per ray), wide in the output features
            return mip::launch_thin_wgrad(M, nout, ncols, dY, ldy, X, ldx, rowdiv, dW, ldw, 1, nullptr, acc, part, st);
        hipError_t e = mip::launch_gemm_f32(true, nout, ncols, M, dY, ldy, X, ldx, rowdiv, false, dW, ldw, acc, splits, part, st);
        if (e == hipSuccess && db) e = mip::launch_gemm_f32(true, nout, 1, M, dY, ldy, nullptr, 0, 1, true, db, 1, acc, splits, part, st);
        return e;
    };
    // dX[M, nin] = dY[M, nout] Wt[nout, ldw] (first nin columns) [+ r1_col[m * r1_ld] * r1_row[n]], then (optionally) the ReLU
    // mask of the layer input x
    auto dgrad = [&](const float* dY, int64_t ldy, int nout, const float* Wt, int64_t ldw, int nin, float* dX, int64_t ldxo,
                     int relu_slot, const float* r1_col = nullptr, int64_t r1_ld = 0, const float* r1_row = nullptr) -> hipError_t {
        const float* relu_x = relu_slot >= 0 ? slot(relu_slot) : nullptr;
        if (mip::gemm_f32_big_ok(Mi, nin, nout, dY, ldy))
            return mip::launch_gemm_f32_big(false, Mi, nin, nout, dY, ldy, Wt, ldw, dX, ldxo, false, 1, nullptr, relu_x, r1_col, r1_ld,
                                            r1_row, nullptr, st, relu_slot >= 0 ? slot_bits(relu_slot) : nullptr);
        hipError_t e = mip::launch_gemm_f32(false, Mi, nin, nout, dY, ldy, Wt, ldw, 1, false, dX, ldxo, false, 1, nullptr, st);
        if (e == hipSuccess && r1_col)
            e = mip::launch_gemm_f32(false, Mi, nin, 1, r1_col, r1_ld, r1_row, nin, 1, false, dX, ldxo, true, 1, nullptr, st);
        if (e == hipSuccess && relu_x) e = mip::launch_relu_mask((int64_t)M * nin, relu_x, dX, st);
        return e;
    };
    if (views) {
        const float* hv = slot(D + Dc);    // output of the LAST view layer [M, Wc] (slot D + 1 + i = output of view layer i)
        const float* bott = slot(D);       // bottleneck [M, W]
        // colour layer (mip_nerf.py:110): d_rgb = d_raw[:, 0:3]
        HIP_TRY(wgrad(d_raw, 4, RGB, hv, Wc, 1, Wc, G(tCW), Wc, G(tCB)));
        // g_hv = (d_rgb Wc) * relu'   [M, Wc] in g0
        HIP_TRY(dgrad(d_raw, 4, RGB, P(tCW), Wc, Wc, g0, Wc, D + Dc));
        // view layers Dc-1 .. 1 (Wc -> Wc, mip_nerf.py:62-69 with net_depth_condition > 1): delta ping-pongs g0 -> g1 -> g0;
        // an odd number of them leaves it in g1: copy back so the first view layer below reads g0 as it always did
        {
            float *ga = g0, *gb = g1;
            for (int i = Dc - 1; i >= 1; --i) {
                HIP_TRY(wgrad(ga, Wc, Wc, slot(D + i), Wc, 1, Wc, G(tVW + 2 * i), Wc, G(tVB + 2 * i)));
                HIP_TRY(dgrad(ga, Wc, Wc, P(tVW + 2 * i), Wc, Wc, gb, Wc, D + i));
                float* t = ga; ga = gb; gb = t;
            }
            if (ga != g0) HIP_TRY(hipMemcpyAsync(g0, ga, (size_t)M * Wc * 4, hipMemcpyDeviceToDevice, st));
        }
        // view layer (mip_nerf.py:106-109): input [bottleneck | view encoding of the sample's ray]
        HIP_TRY(wgrad(g0, Wc, Wc, bott, W, 1, W, G(tVW), W + V, G(tVB)));
        HIP_TRY(wgrad(g0, Wc, Wc, viewenc, 32, N, V, G(tVW) + W, W + V, nullptr));
        // g_bott = g_hv Wv[:, :W]   [M, W] in g1
        HIP_TRY(dgrad(g0, Wc, Wc, P(tVW), W + V, W, g1, W, -1));
        // bottleneck (extra_layer, :102) and density head (:100)
        HIP_TRY(wgrad(g1, W, W, x8, W, 1, W, G(tExW), W, G(tExB)));
        HIP_TRY(wgrad(d_raw + 3, 4, 1, x8, W, 1, W, G(tDensW), W, G(tDensB)));
        // g8 = (g_bott We + d_den Wd) * relu'(x8)   in g0: the density head's dgrad is the rank-1 term of the epilogue
        HIP_TRY(dgrad(g1, W, W, P(tExW), W, W, g0, W, D - 1, d_raw + 3, 4, P(tDensW)));
    } else {
        // MLP.forward(x, None) (mip_nerf.py:99-110): colour and density heads both read the trunk output; extra_layer and
        // view_layers are unused parameters (autograd leaves their .grad None; here: zero unless accumulating)
        HIP_TRY(wgrad(d_raw, 4, RGB, x8, W, 1, W, G(tCW), Wc, G(tCB)));
        HIP_TRY(wgrad(d_raw + 3, 4, 1, x8, W, 1, W, G(tDensW), W, G(tDensB)));
        if (!acc) {
            const int unused[4] = {tExW, tExB, tVW, tVB};
            for (int u = 0; u < 4; ++u)
                HIP_TRY(hipMemsetAsync(G(unused[u]), 0, (size_t)PL.param_numel[unused[u]] * 4, st));
        }
        HIP_TRY(dgrad(d_raw, 4, RGB, P(tCW), Wc, W, g0, W, D - 1, d_raw + 3, 4, P(tDensW)));
    }
    bool d_enc_written = false;
    float* g = g0;          // delta of layer i (gradient w.r.t. its pre-activation)
    float* gn = g1;
    for (int i = D - 1; i >= 0; --i) {
        const int ld = PL.param_numel[2 * i] / W;                    // in_features of layer i
        const float* xin = i == 0 ? enc : slot(i - 1);
        const int nin = i == 0 ? E : W;
        HIP_TRY(wgrad(g, W, W, xin, nin, 1, nin, G(2 * i), ld, G(2 * i + 1)));
        if (ld > nin) HIP_TRY(wgrad(g, W, W, enc, E, 1, E, G(2 * i) + W, ld, nullptr));      // skip concat (:96-97)
        if (d_enc && (ld > nin || i == 0)) {
            // the encoding columns of this layer's weight: [W, W + E) of the skip layer, [0, E) of layer 0
            if (!mip::gemm_f32_big_ok(Mi, E, W, g, W)) return fail(MIPNERF_E_UNSUPPORTED, "mlp_backward_f32_enc: encoding narrower than 64");
            HIP_TRY(mip::launch_gemm_f32_big(false, Mi, E, W, g, W, P(2 * i) + (i == 0 ? 0 : W), ld, d_enc, E, d_enc_written, 1, nullptr,
                                             nullptr, nullptr, 0, nullptr, nullptr, st));
            d_enc_written = true;
        }
        if (i > 0) {
            HIP_TRY(dgrad(g, W, W, P(2 * i), ld, W, gn, W, i - 1));
            float* t = g; g = gn; gn = t;
        }
    }
    return MIPNERF_OK;
}

// ---- the whole training step of the hot path in one call ---------------------------------------------------------
// MipNeRFSystem.training_step (nerf_system.py:95-111) = MipNerf.forward(randomized) + loss, followed by what
// loss.backward() does to the 24 MLP parameters -- native kernels only, no autograd graph, graph-capturable.
// the one-call step covers the bounded model's generated shapes and (round 5) the unbounded-scene model's two-kernel bf16 form
static inline bool has_train_step(const mipnerf_ctx* c) {
    return has_bf16_train(c->P) || (c->cfg.unbounded && has_bf16_train_pre(c->P));
}
size_t mipnerf_train_workspace_bytes(const mipnerf_ctx* c, int64_t B) {
    if (!c || B < 1 || !has_train_step(c)) return 0;
    const size_t N = c->cfg.num_samples, M = (size_t)B * N, L = c->cfg.num_levels;
    size_t act, masks, delta, partials;
    if (mipnerf_mlp_train_sizes(c, (int64_t)M, &act, &masks, &delta, &partials)) return 0;
    size_t per_level = align256(B * (N + 1) * 4) + align256(B * N * 4) + align256(B * 3 * 4) + 2 * align256(B * 4) +   // t, w, rgb, dist, acc
                       align256(((M + 255) / 256) * 256 * c->P->xyz_dim * 2) + 2 * align256(M * 16) +              // enc (whole tiles), rgb_sigma, raw
                       align256(masks) + align256(B * 4) + align256(B * N * 4) + align256(B * 3 * 4);   // masks, ray_loss, d_w, g_rgb
    // act and delta: one contiguous run of wave tiles over ALL levels (one weight-gradient launch over both levels)
    if (c->cfg.unbounded) per_level += align256(B * (N + 1) * 4) + 256;      // inverse-depth fence posts; every level's act region starts aligned
    return 256 + L * per_level + align256(B * 32 * 2) + align256(L * align256(act)) + align256(L * delta) + align256(partials) + align256(M * 16) + 256;
}

int mipnerf_train_step(mipnerf_ctx* c, int64_t B, const mipnerf_rays* rays, const float* gt_rgb, const float* t_rand,
                       const float* u_rand, const float* density_randn, uint32_t flags, float coarse_loss_mult, float distloss_mult,
                       int32_t disable_multiscale_loss, void* workspace, size_t workspace_bytes, float* grad_flat,
                       int32_t accumulate, float* out_scalars, const mipnerf_level_out* out, void* stream) {
    if (!c || !rays || !gt_rgb || !workspace || !grad_flat || !out_scalars || B < 1)
        return fail(MIPNERF_E_INVALID, "train_step: bad argument");
    if (!has_train_step(c))
        return fail(MIPNERF_E_UNSUPPORTED, "train_step: no bf16 training kernels for this variant; it trains in fp32 precision");
    if (c->cfg.unbounded && !has_bf16_train_pre(c->P))      // the unbounded branch hands FRAGMENT encodings to the two-kernel form only
        return fail(MIPNERF_E_UNSUPPORTED, "train_step: the unbounded-scene model trains through the two-kernel (pre-GEMM) form");
    if (!rays->origins || !rays->directions || !rays->viewdirs || !rays->radii || !rays->near || !rays->far || !rays->lossmult)
        return fail(MIPNERF_E_INVALID, "train_step: a Rays field is null");
    if (!c->params_set) return fail(MIPNERF_E_INVALID, "train_step: mipnerf_set_params has not been called");
    if ((t_rand == nullptr) != (u_rand == nullptr) && c->cfg.num_levels > 1)
        return fail(MIPNERF_E_INVALID, "train_step: t_rand and u_rand must both be given (randomized) or both null");
    if (workspace_bytes < mipnerf_train_workspace_bytes(c, B)) return fail(MIPNERF_E_WORKSPACE, "train_step: workspace too small");
    const mipnerf_config& cfg = c->cfg;
    const int N = cfg.num_samples, L = cfg.num_levels;
    const size_t M = (size_t)B * N;
    size_t act_b, mask_b, delta_b, part_b;
    int rc = mipnerf_mlp_train_sizes(c, (int64_t)M, &act_b, &mask_b, &delta_b, &part_b);
    if (rc) return rc;
    TrainWs ws;
    ws.base = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const bool unb = cfg.unbounded != 0;       // inverse-depth fence posts, contracted off-axis IPE, the two-kernel MLP form (mipnerf_forward's unbounded branch)
    struct Lvl { float *t, *t_inv, *w, *rgb, *dist, *acc, *rgb_sigma, *raw, *ray_loss, *d_w, *g_rgb; void *enc, *act, *masks; } lv[2];
    for (int l = 0; l < L; ++l) {
        lv[l].t = ws.take<float>(B * (N + 1) * 4); lv[l].w = ws.take<float>(B * N * 4); lv[l].rgb = ws.take<float>(B * 3 * 4);
        lv[l].t_inv = unb ? ws.take<float>(B * (N + 1) * 4) : nullptr;
        lv[l].dist = ws.take<float>(B * 4); lv[l].acc = ws.take<float>(B * 4);
        lv[l].enc = ws.take<char>(unb ? pre_frag_bytes(c, M) : M * c->P->xyz_dim * 2);      // (fragments cover whole 256-sample tiles)
        lv[l].rgb_sigma = ws.take<float>(M * 16); lv[l].raw = ws.take<float>(M * 16);
        lv[l].masks = ws.take<char>(mask_b);
        lv[l].ray_loss = ws.take<float>(B * 4); lv[l].d_w = ws.take<float>(B * N * 4); lv[l].g_rgb = ws.take<float>(B * 3 * 4);
    }
    void* viewenc = ws.take<char>(B * 32 * 2);
    // T-blocks of the activations and of the deltas: level l owns wave tiles [l * n_wt, (l + 1) * n_wt) of one contiguous run,
    // so that ONE weight-gradient launch (+ one reduction) covers every level (padding tiles carry delta = 0)
    // (two-kernel form: a level's region also holds the encoding record and k_pre_gemm's two outputs behind its T-blocks, and every level
    // gets its own weight-gradient launch -- the kernel reads ONE encoding per launch)
    const size_t act_stride = unb ? align256(act_b) : act_b;
    char* act_all = ws.take<char>((size_t)L * act_stride);
    char* delta_all = ws.take<char>((size_t)L * delta_b);
    for (int l = 0; l < L; ++l) lv[l].act = act_all + (size_t)l * act_stride;
    float* partials = ws.take<float>(part_b);
    float* d_raw = ws.take<float>(M * 16);
    const int disparity = (cfg.disparity || (flags & MIPNERF_FLAG_DISPARITY)) ? 1 : 0;
    const int white = (flags & MIPNERF_FLAG_WHITE_BKGD) ? 1 : 0;
    // ---- forward (mip_nerf.py:182-246), activations saved for the backward -------------------------------------------
    // option 4 (fuse_small): pos_enc + coarse fence posts in one launch; per level compositing + distloss (+ the next level's
    // fence posts) in one launch instead of three -- same per-ray device functions, same bits
    const bool fuse_tail = c->fuse_small != 0;        // every N <= MIPNERF_MAX_SAMPLES (round 5: one set of K buckets in all per-ray kernels)
    if (c->fuse_small && !unb) {
        HIP_TRY(mip::launch_ray_prologue(B, cfg.deg_view, rays->viewdirs, viewenc, 32, true, N, rays->near, rays->far, t_rand, disparity,
                                         lv[0].t, S(stream)));
    } else if ((rc = mipnerf_pos_enc(B, cfg.deg_view, rays->viewdirs, viewenc, 32, MIPNERF_PREC_BF16, stream))) {
        return rc;
    }
    for (int l = 0; l < L; ++l) {
        const float* dnoise = density_randn ? density_randn + (size_t)l * M : nullptr;      // this level's draws (mip_nerf.py:232-233)
        if (unb) {
            // the fine level inverts the coarse weights' PDF over the INVERSE-DEPTH fence posts, then t = 1 / t_inv (as mipnerf_forward)
            if (l == 0) {
                HIP_TRY(mip::launch_sample_along_rays_360(B, N, rays->near, rays->far, t_rand, lv[0].t_inv, lv[0].t, S(stream)));
            } else {
                if (!fuse_tail && (rc = mipnerf_resample_along_rays(B, N, lv[l - 1].t_inv, lv[l - 1].w, u_rand, cfg.resample_padding,
                                                                    lv[l].t_inv, stream))) return rc;
                HIP_TRY(mip::launch_reciprocal((int64_t)B * (N + 1), lv[l].t_inv, lv[l].t, S(stream)));
            }
        } else if (l == 0) {
            if (!c->fuse_small && (rc = mipnerf_sample_along_rays(B, N, rays->near, rays->far, t_rand, disparity, lv[0].t, stream))) return rc;
        } else if (!fuse_tail) {
            if ((rc = mipnerf_resample_along_rays(B, N, lv[l - 1].t, lv[l - 1].w, u_rand, cfg.resample_padding, lv[l].t, stream))) return rc;
        }
        if (unb) {      // the 672-wide encoding as bf16 B-operand fragments: k_pre_gemm reads them, and so does this level's weight-gradient launch
            HIP_TRY(mip::launch_cast_ipe_360(B, N, cfg.min_deg_point, cfg.max_deg_point, 1, lv[l].t, rays->origins, rays->directions, rays->radii,
                                             lv[l].enc, true, nullptr, nullptr, S(stream), true));
            if ((rc = mlp_forward_train_noise(c, (int64_t)M, N, lv[l].enc, viewenc, lv[l].rgb_sigma, lv[l].raw, lv[l].act, lv[l].masks, dnoise,
                                              stream, true))) return rc;
        } else if (c->fused_ipe && max_deg_span_is_16(cfg)) {      // encoding computed inside the forward-with-save kernel
            const mip::RayInputs ri = {lv[l].t, rays->origins, rays->directions, rays->radii, cfg.min_deg_point,
                                       cfg.disable_integration};
            HIP_TRY(launch_trainfwd_variant(c, nullptr, viewenc, lv[l].rgb_sigma, lv[l].raw, lv[l].act, lv[l].masks, (int64_t)M, N, &ri,
                                            dnoise, S(stream)));
        } else {
            if ((rc = mipnerf_cast_ipe(B, N, cfg.min_deg_point, cfg.max_deg_point, cfg.disable_integration, lv[l].t, rays->origins,
                                       rays->directions, rays->radii, lv[l].enc, MIPNERF_PREC_BF16, stream))) return rc;
            if ((rc = mlp_forward_train_noise(c, (int64_t)M, N, lv[l].enc, viewenc, lv[l].rgb_sigma, lv[l].raw, lv[l].act,
                                              lv[l].masks, dnoise, stream))) return rc;
        }
        // distloss (mip.py:8-20) forward AND backward in one pass: d loss / d ray_loss is the constant k_l * dm / B
        const float k = (L > 1 && l == 0) ? coarse_loss_mult : 1.0f;
        if (fuse_tail) {
            HIP_TRY(mip::launch_composite_train(B, N, lv[l].rgb_sigma, lv[l].t, rays->directions, white, lv[l].rgb, lv[l].dist, lv[l].acc,
                                                lv[l].w, lv[l].ray_loss, k * distloss_mult / (float)B, lv[l].d_w, u_rand,
                                                cfg.resample_padding, l + 1 < L ? (unb ? lv[l + 1].t_inv : lv[l + 1].t) : nullptr, S(stream),
                                                unb ? lv[l].t_inv : nullptr));
            continue;
        }
        if ((rc = mipnerf_volumetric_rendering(B, N, lv[l].rgb_sigma, lv[l].t, rays->directions, white, lv[l].rgb, lv[l].dist,
                                               lv[l].acc, lv[l].w, stream))) return rc;
        HIP_TRY(mip::launch_distloss(B, N, lv[l].w, lv[l].t, lv[l].ray_loss, nullptr, k * distloss_mult / (float)B, lv[l].d_w,
                                     S(stream)));
    }
    // ---- loss (nerf_system.py:99-111) and d loss / d comp_rgb ------------------------------------------------------------
    HIP_TRY(mip::launch_loss_fused(B, L, lv[0].rgb, L > 1 ? lv[1].rgb : nullptr, gt_rgb, disable_multiscale_loss ? nullptr : rays->lossmult,
                                   lv[0].ray_loss, L > 1 ? lv[1].ray_loss : nullptr, coarse_loss_mult, distloss_mult, lv[0].g_rgb,
                                   L > 1 ? lv[1].g_rgb : lv[0].g_rgb, out_scalars, S(stream)));
    // ---- backward: compositing + activations, then the MLP, level by level (resampling carries no gradient,
    //      mip_nerf.py:187-196 stop_resample_grad) ----------------------------------------------------------------------------
    for (int l = L - 1; l >= 0; --l) {
        if ((rc = mipnerf_volumetric_rendering_bwd(B, N, lv[l].rgb_sigma, lv[l].t, rays->directions, white, lv[l].g_rgb, nullptr,
                                                   nullptr, lv[l].d_w, cfg.rgb_padding, d_raw, stream))) return rc;
        if ((rc = mipnerf_mlp_dgrad(c, (int64_t)M, d_raw, lv[l].masks, delta_all + (size_t)l * delta_b, stream))) return rc;
        if (unb && (rc = wgrad_tiles(c, (int64_t)(((M + 255) / 256) * 8), lv[l].act, delta_all + (size_t)l * delta_b, partials, grad_flat,
                                     (accumulate || l < L - 1) ? 1 : 0, stream))) return rc;
    }
    // one weight-gradient pass over the wave tiles of all levels (the sum over levels is part of the sample contraction)
    if (!unb && (rc = wgrad_tiles(c, (int64_t)L * (int64_t)(((M + 255) / 256) * 8), act_all, delta_all, partials, grad_flat, accumulate ? 1 : 0,
                                  stream))) return rc;
    if (out)      // optional copies of what MipNerf.forward returns (async device-to-device)
        for (int l = 0; l < L; ++l) {
            const mipnerf_level_out& o = out[l];
            if (o.comp_rgb) HIP_TRY(hipMemcpyAsync(o.comp_rgb, lv[l].rgb, B * 3 * 4, hipMemcpyDeviceToDevice, S(stream)));
            if (o.distance) HIP_TRY(hipMemcpyAsync(o.distance, lv[l].dist, B * 4, hipMemcpyDeviceToDevice, S(stream)));
            if (o.acc) HIP_TRY(hipMemcpyAsync(o.acc, lv[l].acc, B * 4, hipMemcpyDeviceToDevice, S(stream)));
            if (o.weights) HIP_TRY(hipMemcpyAsync(o.weights, lv[l].w, B * N * 4, hipMemcpyDeviceToDevice, S(stream)));
            if (o.t_samples) HIP_TRY(hipMemcpyAsync(o.t_samples, lv[l].t, B * (N + 1) * 4, hipMemcpyDeviceToDevice, S(stream)));
        }
    return MIPNERF_OK;
}

// ---- the level loop ---------------------------------------------------------------------------------
// encoding region of one level: [M, xyz_dim] fp32 (or bf16) -- or, two-kernel bf16 form, the bf16 fragments of whole 256-sample tiles followed by
// k_pre_gemm's two outputs (1,344 + 1,536 B per sample: 7 % more than the fp32 encodings they replace, so one region serves either precision)
static size_t enc_region_bytes(const mipnerf_ctx* c, size_t M) {
    const size_t rowmajor = M * c->P->xyz_dim * 4;
    const size_t pre = has_bf16_pre(c->P) ? align256(pre_frag_bytes(c, M)) + align256(pre_x_bytes((int64_t)M)) + align256(pre_acc_bytes((int64_t)M)) : 0;
    return rowmajor > pre ? rowmajor : pre;
}
size_t mipnerf_workspace_bytes(const mipnerf_ctx* c, int64_t B) {
    if (!c || B < 1) return 0;
    const size_t M = (size_t)B * (size_t)c->cfg.num_samples;
    return align256(enc_region_bytes(c, M)) + align256((size_t)B * 32 * 4) + align256(M * 16) + 256 +
           (c->cfg.unbounded ? 2 * align256((size_t)B * (c->cfg.num_samples + 1) * 4) : 0);       // inverse-depth fence posts of two levels
}

int mipnerf_forward(mipnerf_ctx* c, int64_t B, const mipnerf_rays* rays, const float* t_rand, const float* u_rand,
                    const float* density_randn, uint32_t flags, int precision, void* workspace, size_t workspace_bytes,
                    const mipnerf_level_out* out, void* stream) {
    if (!c || !rays || !out || !workspace || B < 1) return fail(MIPNERF_E_INVALID, "forward: bad argument");
    if (!rays->origins || !rays->directions || !rays->viewdirs || !rays->radii || !rays->near || !rays->far)
        return fail(MIPNERF_E_INVALID, "forward: a Rays field is null");
    if (!c->params_set) return fail(MIPNERF_E_INVALID, "forward: mipnerf_set_params has not been called");
    if ((t_rand == nullptr) != (u_rand == nullptr) && c->cfg.num_levels > 1)
        return fail(MIPNERF_E_INVALID, "forward: t_rand and u_rand must both be given (randomized) or both null");
    if (workspace_bytes < mipnerf_workspace_bytes(c, B)) return fail(MIPNERF_E_WORKSPACE, "forward: workspace too small");
    if (precision != MIPNERF_PREC_BF16 && precision != MIPNERF_PREC_FP32)
        return fail(MIPNERF_E_INVALID, "unknown precision %d", precision);
    if (precision == MIPNERF_PREC_BF16 && !has_bf16(c->P))
        return fail(MIPNERF_E_UNSUPPORTED, "forward: this architecture variant has fp32 kernels only (csrc/gen_mlp_bf16.py VARIANTS)");
    const mipnerf_config& cfg = c->cfg;
    const int N = cfg.num_samples;
    const size_t M = (size_t)B * N;
    char* ws = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    void* enc = ws;
    const size_t enc_b = align256(enc_region_bytes(c, M));
    void* viewenc = ws + enc_b;
    float* rgb_sigma = reinterpret_cast<float*>(ws + enc_b + align256((size_t)B * 32 * 4));
    float* t_inv[2] = {nullptr, nullptr};
    if (cfg.unbounded) {
        char* q = reinterpret_cast<char*>(rgb_sigma) + align256(M * 16);
        t_inv[0] = reinterpret_cast<float*>(q);
        t_inv[1] = reinterpret_cast<float*>(q + align256((size_t)B * (N + 1) * 4));
    }
    // bf16 on a variant whose encoding is too wide for k_mlp_bf16 (the unbounded-scene model): k_pre_gemm + trunk kernel (gen_pre_gemm.py);
    // its fragments and the two buffers between the kernels share the encoding region
    const bool pre_form = precision == MIPNERF_PREC_BF16 && has_bf16_pre(c->P);
    char* pre_x = pre_form ? ws + align256(pre_frag_bytes(c, M)) : nullptr;
    char* pre_acc = pre_form ? pre_x + align256(pre_x_bytes((int64_t)M)) : nullptr;
    const int disparity = (cfg.disparity || (flags & MIPNERF_FLAG_DISPARITY)) ? 1 : 0;
    const int white = (flags & MIPNERF_FLAG_WHITE_BKGD) ? 1 : 0;
    int rc;
    // pos_enc(viewdirs) is level-independent: computed once (the reference recomputes it, mip_nerf.py:220-226) -- in the same launch
    // as the coarse level's fence posts (option 4 = 0: one launch per stage, the per-stage kernels; both routes give the same bits)
    bool have_t0 = false;
    if (c->fuse_small && !cfg.unbounded) {
        if (!out[0].t_samples) return fail(MIPNERF_E_INVALID, "forward: output pointer of level 0 is null");
        HIP_TRY(mip::launch_ray_prologue(B, cfg.deg_view, rays->viewdirs, viewenc, 32, precision == MIPNERF_PREC_BF16, N, rays->near, rays->far,
                                         t_rand, disparity, out[0].t_samples, S(stream)));
        have_t0 = true;
    } else if ((rc = mipnerf_pos_enc(B, cfg.deg_view, rays->viewdirs, viewenc, 32, precision, stream))) {
        return rc;
    }
    bool have_resampled = false;      // the previous level's launch already drew this level's fence posts (k_composite_resample)
    for (int lvl = 0; lvl < cfg.num_levels; ++lvl) {
        const float* dnoise = density_randn ? density_randn + (size_t)lvl * M : nullptr;    // this level's draws (mip_nerf.py:232-233)
        const mipnerf_level_out& o = out[lvl];
        if (!o.comp_rgb || !o.distance || !o.acc || !o.weights || !o.t_samples)
            return fail(MIPNERF_E_INVALID, "forward: output pointer of level %d is null", lvl);
        const bool fused = !cfg.unbounded && precision == MIPNERF_PREC_BF16 && c->fused_ipe && c->mlp_dma;
        if (cfg.unbounded) {
            // SURVEY 8(f)-4 (what mip.py:106-124, 292-319, 424-447 aim at): fence posts uniform in inverse depth; the fine level
            // inverts the coarse weights' piecewise-constant PDF over the INVERSE-DEPTH fence posts (the same blur pool + padding
            // as mip.py:252-257; the inversion only interpolates between bins, so it is the s-space resampling of the paper up
            // to the affine map s <-> 1/t), then t = 1 / t_inv; contracted full-covariance Gaussians -> off-axis IPE
            if (lvl == 0) {
                HIP_TRY(mip::launch_sample_along_rays_360(B, N, rays->near, rays->far, t_rand, t_inv[0], o.t_samples, S(stream)));
            } else {
                if (!have_resampled &&
                    (rc = mipnerf_resample_along_rays(B, N, t_inv[lvl - 1], out[lvl - 1].weights, u_rand, cfg.resample_padding,
                                                      t_inv[lvl], stream))) return rc;
                HIP_TRY(mip::launch_reciprocal((int64_t)B * (N + 1), t_inv[lvl], o.t_samples, S(stream)));
            }
            // dtype of the encoding = the precision (fragments when the two-kernel form will read them); an unbounded variant with bf16
            // kernels of some OTHER form must not be handed fp32 rows here
            if (precision == MIPNERF_PREC_BF16 && !pre_form)
                return fail(MIPNERF_E_UNSUPPORTED, "forward: bf16 inference of an unbounded variant needs the two-kernel (pre-GEMM) form");
            HIP_TRY(mip::launch_cast_ipe_360(B, N, cfg.min_deg_point, cfg.max_deg_point, 1, o.t_samples, rays->origins, rays->directions,
                                             rays->radii, enc, precision == MIPNERF_PREC_BF16, nullptr, nullptr, S(stream), pre_form));
        } else if (lvl == 0) {
            if (!have_t0 && (rc = mipnerf_sample_along_rays(B, N, rays->near, rays->far, t_rand, disparity, o.t_samples, stream))) return rc;
        } else if (!have_resampled) {
            if ((rc = mipnerf_resample_along_rays(B, N, out[lvl - 1].t_samples, out[lvl - 1].weights, u_rand,
                                                  cfg.resample_padding, o.t_samples, stream))) return rc;
        }
        have_resampled = false;
        if (!fused && !cfg.unbounded &&
            (rc = mipnerf_cast_ipe(B, N, cfg.min_deg_point, cfg.max_deg_point, cfg.disable_integration, o.t_samples,
                                   rays->origins, rays->directions, rays->radii, enc, precision, stream))) return rc;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (c->time_mlp == 1) {
            if (c->ev_used + 2 > c->ev.size()) {
                for (int i = 0; i < 64; ++i) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); c->ev.push_back(e); }
            }
            e0 = c->ev[c->ev_used]; e1 = c->ev[c->ev_used + 1]; c->ev_used += 2;
            HIP_TRY(hipEventRecord(e0, S(stream)));
        }
        if (fused) {
            if (max_deg_span_is_16(cfg) == false)
                return fail(MIPNERF_E_UNSUPPORTED, "fused IPE is generated for max_deg-min_deg == 16");
            const mip::RayInputs ri = {o.t_samples, rays->origins, rays->directions, rays->radii, cfg.min_deg_point,
                                       cfg.disable_integration};
            HIP_TRY(launch_bf16_variant(c, nullptr, viewenc, rgb_sigma, nullptr, (int64_t)M, N, true, &ri, dnoise, S(stream)));
        } else if (pre_form) {
            HIP_TRY(launch_bf16_pre(c, enc, 1, viewenc, rgb_sigma, nullptr, (int64_t)M, N, pre_x, pre_acc, dnoise, S(stream)));
        } else if ((rc = mlp_forward_noise(c, (int64_t)M, N, enc, viewenc, precision, rgb_sigma, nullptr, dnoise, stream))) {
            return rc;
        }
        if (c->time_mlp == 1) HIP_TRY(hipEventRecord(e1, S(stream)));
        if (c->fuse_small && lvl + 1 < cfg.num_levels) {
            // compositing of this level + the next level's fence posts in one launch (weights go registers -> LDS, not through HBM)
            const mipnerf_level_out& nx = out[lvl + 1];
            if (!nx.t_samples) return fail(MIPNERF_E_INVALID, "forward: output pointer of level %d is null", lvl + 1);
            HIP_TRY(mip::launch_composite_resample(B, N, rgb_sigma, o.t_samples, rays->directions, white, o.comp_rgb, o.distance, o.acc,
                                                   o.weights, cfg.unbounded ? t_inv[lvl] : o.t_samples, u_rand, cfg.resample_padding,
                                                   cfg.unbounded ? t_inv[lvl + 1] : nx.t_samples, S(stream)));
            have_resampled = true;
            continue;
        }
        if ((rc = mipnerf_volumetric_rendering(B, N, rgb_sigma, o.t_samples, rays->directions, white, o.comp_rgb,
                                               o.distance, o.acc, o.weights, stream))) return rc;
    }
    return MIPNERF_OK;
}

// ---- instrumentation -----------------------------------------------------------------------------------
int mipnerf_time_mlp(mipnerf_ctx* c, int64_t M, int32_t N, const void* enc, const void* viewenc, int precision,
                     float* rgb_sigma, int iters, float* ms, void* stream) {
    if (!ms || iters < 1) return fail(MIPNERF_E_INVALID, "time_mlp: bad argument");
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    int rc = mipnerf_mlp_forward(c, M, N, enc, viewenc, precision, rgb_sigma, nullptr, stream);   // warm
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e0, S(stream)));
    for (int i = 0; i < iters; ++i)
        if ((rc = mipnerf_mlp_forward(c, M, N, enc, viewenc, precision, rgb_sigma, nullptr, stream))) return rc;
    HIP_TRY(hipEventRecord(e1, S(stream)));
    HIP_TRY(hipEventSynchronize(e1));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = t / iters;
    return MIPNERF_OK;
}

int mipnerf_mlp_launch_stats(mipnerf_ctx* c, double* total_ms, int64_t* launches) {
    if (!c || !total_ms || !launches) return fail(MIPNERF_E_INVALID, "mlp_launch_stats: null argument");
    double tot = 0;
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(c->ev[i + 1]));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1]));
        tot += t;
    }
    *total_ms = tot;
    *launches = (int64_t)(c->ev_used / 2);
    c->ev_used = 0;
    return MIPNERF_OK;
}

int mipnerf_selftest(void* stream) {
    char msg[256];
    const int bad = mip::run_selftest(S(stream), msg, sizeof msg);
    g_err = msg;
    if (bad < 0) return MIPNERF_E_HIP;
    return bad == 0 ? MIPNERF_OK : (0x100 | bad);
}

// Host-only debug export of the plan tables (flat parameter indices), used by the CPU tests to prove the
// C++ expansion equals mlp_plan.py.  which: 0 bf16 pack, 1 bias, 2 fp32 pack.  Returns element count.
int64_t mipnerf_debug_table_variant(int variant, int which, int32_t* out_host, int64_t cap) {
    if (variant < 0 || variant >= mip::plan::kNumVariants || which < 0 || which > 5) return -1;
    if (which >= 3) {               // training tables of the variant (3 dgrad pack, 4 wgrad partial -> parameter, 5 jobs)
        TrainTables tt;
        if (!train_tables(tt, variant)) return -1;
        const int32_t* src = which == 3 ? tt.bpack : (which == 4 ? tt.otab : tt.jobs);
        const int64_t n = which == 3 ? (int64_t)tt.n_bchunks * 512 : (which == 4 ? (int64_t)tt.njobs * tt.job_floats : tt.njobs * 20);
        if (out_host && cap >= n) memcpy(out_host, src, (size_t)n * 4);
        return n;
    }
    Tables T;
    build_tables(T, mip::plan::kPlans[variant]);
    const std::vector<int32_t>& v = which == 0 ? T.pack_bf16 : (which == 1 ? T.bias : T.pack_f32);
    if (out_host && cap >= (int64_t)v.size()) memcpy(out_host, v.data(), v.size() * 4);
    return (int64_t)v.size();
}

int64_t mipnerf_debug_table(int which, int32_t* out_host, int64_t cap) {
    if (which >= 3 && which <= 5) {
        TrainTables tt;
        if (!train_tables(tt)) return -1;
        const int32_t* src = which == 3 ? tt.bpack : (which == 4 ? tt.otab : tt.jobs);
        const int64_t n = which == 3 ? (int64_t)tt.n_bchunks * 512 : (which == 4 ? (int64_t)tt.njobs * tt.job_floats : tt.njobs * 20);
        if (out_host && cap >= n) memcpy(out_host, src, (size_t)n * 4);
        return n;
    }
    Tables T;
    build_tables(T, mip::plan::kPlans[0]);
    const std::vector<int32_t>& v = which == 0 ? T.pack_bf16 : (which == 1 ? T.bias : T.pack_f32);
    if (out_host && cap >= (int64_t)v.size()) memcpy(out_host, v.data(), v.size() * 4);
    return (int64_t)v.size();
}

// Host-only: fp32 layer descriptors (x_in0, kb0, x_in1, kb1, x_out, ntiles, first_tile, relu, kind, chunk0, stage_view, ldx) x nlayers
int64_t mipnerf_debug_f32net(int32_t* out_host, int64_t cap) {
    Tables T;
    build_tables(T, mip::plan::kPlans[0]);
    const int n = T.net.nlayers;
    if (out_host && cap >= (int64_t)n * 12)
        for (int i = 0; i < n; ++i) {
            const mip::F32Layer& L = T.net.layers[i];
            const int32_t row[12] = {L.x_in0, L.kb0, L.x_in1, L.kb1, L.x_out, L.ntiles, L.first_tile, L.relu, L.kind, L.chunk0,
                                     L.stage_view, T.net.ldx};
            memcpy(out_host + i * 12, row, sizeof row);
        }
    return n;
}

}  // extern "C"
