// Network-level entry points: ResNet-101 trunk to block3 (output stride 4), squash + map decoder + xyz-map head,
// centroid / shape regression heads.  Host-side layer sequencing in C++ over the kernels of conv_mfma.hip and
// image_ops.hip; no allocation, no synchronisation, everything on the caller's stream, scratch carved from the
// caller's workspace.
#include <atomic>
#include <vector>

#include "common.h"
#include "wino3_filter.h"

namespace mpsr {
int conv2d(const float *x, int B, int H, int W, int C, const float *w, const float *bias, const float *residual,
           float *y, int N, int KH, int KW, int dilation, int relu, int split_k, float *ws, size_t ws_floats,
           hipStream_t stream);
size_t conv_scratch_floats(long long M, int N);
bool conv2d_takes_winograd4(int B, int H, int W, int C, int N, const float *ws, size_t ws_floats);
int conv3x3_winograd4(const float *x, int B, int H, int W, int C, const float *w, const float *bias, int relu,
                      float *y, int N, float *ws, size_t ws_floats, hipStream_t s, int in_c8, int out_c8, float *part,
                      size_t part_floats);
size_t winograd4_split_floats(int B, int H, int W, int N);
extern std::atomic<int> g_wino4_split;
int winograd3_filter_form(int B, int H, int W, int C, int N, int dilation);  // winograd3.hip
bool conv2d_takes_winograd3(int B, int H, int W, int C, int N, int KH, int KW, int dilation, int split_k, const float *ws,
                            size_t ws_floats);
bool conv2d_takes_pointwise(long long M, int C, int N, int KH, int KW, int split_k);
int conv3x3_narrow(const float *x, int B, int H, int W, int C, const float *w, const float *bias, int relu, float *y,
                   int N, hipStream_t s, int in_c8);
bool conv3x3_narrow_takes_mfma(const float *x, int B, int H, int W, int C, int N, const float *w);
int conv2d_winograd_choice(int B, int H, int W, int C, int N, int KH, int KW, int dilation, bool residual, int split_k,
                           const float *ws, size_t ws_floats);
int resize_bilinear_c8(const float *in, int B, int H, int W, int C, int OH, int OW, int align_corners, float *out,
                       hipStream_t s);
// upconv.hip: 3x3 convolution of a bilinearly upsampled map as a low-resolution tap GEMM + gather
bool upconv_applies(int B, int h, int w, int C, int OH, int OW, int N, int align_corners);
size_t upconv_z_floats(long long Msrc, int N);
size_t upconv_weight_floats(int C, int N);
int conv3x3_upsampled(const float *x, int B, int h, int w, int C, int OH, int OW, int align_corners, const float *g,
                      const float *bias, int relu, float *y, int N, int out_c8, float *z, size_t z_floats, float *ws,
                      size_t ws_floats, hipStream_t s);
}

#include <atomic>
// 1 (default): the map decoder's internal tensors are channel-blocked whenever all four of its 3x3 layers go to the
// F(4x4,3x3) kernel; 0: NHWC throughout (tests compare the two bit for bit)
static std::atomic<int> g_decoder_c8{1};
extern "C" void mpsr_debug_set_decoder_c8(int on) { g_decoder_c8 = on; }
// 1 (default): the decoder's two convolutions that follow an upsampling (conv2_1, conv3_1) run as a tap GEMM on the
// source map + gather (upconv.hip) inside the channel-blocked chain; 0: resize + F(4x4,3x3) as in round 3
static std::atomic<int> g_decoder_upconv{1};
extern "C" void mpsr_debug_set_decoder_upconv(int on) { g_decoder_upconv = on; }

namespace {

// bump allocator over the caller's workspace; 256-byte aligned pieces
struct Arena {
    char *base;
    size_t size, used = 0;
    bool ok = true;
    Arena(void *p, size_t n) : base(static_cast<char *>(p)), size(n) {}
    float *floats(size_t n)
    {
        const size_t bytes = mpsr::align_up(n * sizeof(float), 256);
        if (used + bytes > size) {
            ok = false;
            return nullptr;
        }
        float *r = reinterpret_cast<float *>(base + used);
        used += bytes;
        return r;
    }
};

inline size_t fbytes(size_t n) { return mpsr::align_up(n * sizeof(float), 256); }

int run_layer(const float *blob, const mpsr_layer &L, const float *x, int B, int H, int W, const float *residual,
              float *y, int split_k, float *ws, size_t ws_floats, hipStream_t s)
{
    return mpsr::conv2d(x, B, H, W, L.cin, blob + L.w_off, L.b_off >= 0 ? blob + L.b_off : nullptr, residual, y, L.cout,
                        L.kh, L.kw, L.dilation, L.relu, split_k, ws, ws_floats, s);
}

constexpr int kTrunkUnits[3] = {3, 4, 23};

// floats of filter cache a 3x3 layer gets: F(4x4,3x3) keeps 36 values per filter, F(3x3,3x3) (atrous layers) 25
inline size_t cache_floats_of(const mpsr_layer &L)
{
    if (L.kh != 3 || L.kw != 3) return 0;
    return mpsr::align_up((size_t)(L.dilation > 1 ? 25 : 36) * (size_t)L.cout * (size_t)L.cin, 64);
}

// Walks the caller's filter cache (mpsr_net_opts) layer by layer in execution order: every 3x3 layer owns a fixed slice,
// offered to the Winograd entry points through g_filter_cache_slot right before the layer runs.
struct FilterCache {
    float *base = nullptr;
    size_t floats = 0, used = 0;
    bool valid = false;
    int *tags = nullptr;  // the caller's per-layer notes (mpsr_net_opts.filter_cache_tags), indexed by layer record
    explicit FilterCache(const mpsr_net_opts *o)
    {
        if (o && o->filter_cache && o->filter_cache_floats) {
            base = o->filter_cache;
            floats = o->filter_cache_floats;
            valid = o->filter_cache_valid != 0;
            tags = o->filter_cache_tags;
        }
    }
    // slice of layer L (nullptr: no cache, or it is full); call once per 3x3 layer, in order
    float *take(const mpsr_layer &L)
    {
        const size_t need = cache_floats_of(L);
        if (!base || need == 0) return nullptr;
        float *r = used + need <= floats ? base + used : nullptr;
        used += need;
        return r;
    }
    void offer(const float *w, float *slice, const mpsr_layer &L, int layer_index) const
    {
        mpsr::g_filter_cache_slot = mpsr::FilterCacheSlot();
        if (slice) {
            mpsr::g_filter_cache_slot.w = w;
            mpsr::g_filter_cache_slot.u = slice;
            mpsr::g_filter_cache_slot.floats = cache_floats_of(L);
            mpsr::g_filter_cache_slot.ready = valid;
            mpsr::g_filter_cache_slot.tag = tags ? tags + layer_index : nullptr;
        }
    }
};

struct TrunkDims {
    int OH, OW, PH, PW;
};
inline TrunkDims trunk_dims(int H, int W)
{
    TrunkDims d;
    d.OH = (H + 6 - 7) / 2 + 1;
    d.OW = (W + 6 - 7) / 2 + 1;
    d.PH = (d.OH + 1) / 2;
    d.PW = (d.OW + 1) / 2;
    return d;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ trunk

extern "C" size_t mpsr_trunk_workspace_bytes(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const TrunkDims d = trunk_dims(H, W);
    const size_t Mr = (size_t)B * d.OH * d.OW, Mp = (size_t)B * d.PH * d.PW;
    // cols, root, pooled, 3 x (Mp x 1024) ping/pong/shortcut, 2 x (Mp x 256) bottleneck temporaries
    // + the scheduling scratch of the convolution launches (stream-K partial tiles and counters, conv_mfma.hip)
    return fbytes(Mr * 160) + fbytes(Mr * 64) + fbytes(Mp * 64) + 3 * fbytes(Mp * 1024) + 2 * fbytes(Mp * 256) +
           fbytes(mpsr::conv_scratch_floats((long long)(Mr > Mp ? Mr : Mp), 1024));
}

extern "C" size_t mpsr_filter_cache_floats(const mpsr_layer *layers, int n_layers)
{
    size_t n = 0;
    if (layers)
        for (int i = 0; i < n_layers; ++i) n += cache_floats_of(layers[i]);
    return n;
}

extern "C" int mpsr_trunk_fwd(const float *img, int B, int H, int W, const float *blob, const mpsr_layer *layers,
                              int n_layers, float *out, void *workspace, size_t workspace_bytes, mpsr_stream_t stream)
{
    return mpsr_trunk_fwd_ex(img, B, H, W, blob, layers, n_layers, out, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int mpsr_trunk_fwd_ex(const float *img, int B, int H, int W, const float *blob, const mpsr_layer *layers,
                                 int n_layers, float *out, void *workspace, size_t workspace_bytes,
                                 const mpsr_net_opts *opts, mpsr_stream_t stream)
{
    MPSR_REQUIRE(B >= 0 && H >= 7 && W >= 7, "trunk_fwd: bad input shape (B=%d H=%d W=%d)", B, H, W);
    if (opts) {
        MPSR_REQUIRE(opts->math >= 0 && opts->math <= MPSR_CALL_MATH_BF16X3, "%s: unknown opts.math %d", "trunk_fwd", opts->math);
        MPSR_REQUIRE(opts->winograd_policy >= 0 && opts->winograd_policy <= MPSR_CALL_WINOGRAD_ACCURATE,
                     "%s: unknown opts.winograd_policy %d", "trunk_fwd", opts->winograd_policy);
    }
    mpsr::CallOptsGuard call_opts(opts ? opts->math : 0, opts ? opts->winograd_policy : 0);
    MPSR_REQUIRE(n_layers == MPSR_TRUNK_LAYERS, "trunk_fwd: expected %d layer records, got %d", MPSR_TRUNK_LAYERS,
                 n_layers);
    if (B == 0) return MPSR_OK;
    MPSR_REQUIRE(img && blob && layers && out && workspace, "trunk_fwd: null pointer");
    hipStream_t s = mpsr::as_stream(stream);
    const TrunkDims d = trunk_dims(H, W);
    const size_t Mr = (size_t)B * d.OH * d.OW, Mp = (size_t)B * d.PH * d.PW;
    const mpsr_layer &root = layers[0];
    MPSR_REQUIRE(root.kh == 1 && root.kw == 1 && root.cin >= 147 && root.cin % 32 == 0,
                 "trunk_fwd: layer 0 must be the root as a 1x1 layer over im2col columns (cin=%d)", root.cin);
    // widest tensors (narrow test copies of the graph scale with the records)
    int cmax = 0, bmax = 0;
    for (int i = 1; i < n_layers; ++i) {
        if (layers[i].kh == 3 && layers[i].cout > bmax) bmax = layers[i].cout;
        if (layers[i].cout > cmax) cmax = layers[i].cout;
    }
    Arena ar(workspace, workspace_bytes);
    float *cols = ar.floats(Mr * root.cin);
    float *rootout = ar.floats(Mr * root.cout);
    float *pooled = ar.floats(Mp * root.cout);
    float *ping = ar.floats(Mp * cmax), *pong = ar.floats(Mp * cmax), *scut = ar.floats(Mp * cmax);
    float *t1 = ar.floats(Mp * bmax), *t2 = ar.floats(Mp * bmax);
    const size_t skn = mpsr::conv_scratch_floats((long long)(Mr > Mp ? Mr : Mp), cmax > 1024 ? cmax : 1024);
    float *sk = ar.floats(skn);
    if (!ar.ok)
        return mpsr::fail(MPSR_ERR_WORKSPACE, "trunk_fwd: workspace %zu bytes too small (see mpsr_trunk_workspace_bytes)",
                          workspace_bytes);

    int rc;
    struct TailJobGuard {  // no filter-transform job (wino3_filter.h) outlives this call, whichever way it returns
        ~TailJobGuard()
        {
            mpsr::g_filter_tail_job = mpsr::FilterTailJob();
            mpsr::g_filter_tail_done = mpsr::FilterTailJob();
            mpsr::g_filter_cache_slot = mpsr::FilterCacheSlot();
        }
    } tail_job_guard;
    FilterCache cache(opts);
    if ((rc = mpsr_im2col_root(img, B, H, W, cols, root.cin, stream))) return rc;
    if ((rc = run_layer(blob, root, cols, B, d.OH, d.OW, nullptr, rootout, 0, sk, skn, s))) return rc;
    if ((rc = mpsr_max_pool(rootout, B, d.OH, d.OW, root.cout, 3, 2, 1, pooled, stream))) return rc;

    const float *cur = pooled;
    int cur_c = root.cout;
    int li = 1;
    for (int blk = 0; blk < 3; ++blk) {
        for (int u = 0; u < kTrunkUnits[blk]; ++u) {
            const bool last = (blk == 2 && u == kTrunkUnits[blk] - 1);
            const float *residual = cur;
            if (u == 0) {  // projection shortcut: 1x1, BN folded, no activation
                const mpsr_layer &S = layers[li++];
                MPSR_REQUIRE(S.kh == 1 && S.cin == cur_c && !S.relu, "trunk_fwd: record %d is not a shortcut", li - 1);
                if ((rc = run_layer(blob, S, cur, B, d.PH, d.PW, nullptr, scut, 0, sk, skn, s))) return rc;
                residual = scut;
            }
            const mpsr_layer &c1 = layers[li], &c2 = layers[li + 1], &c3 = layers[li + 2];
            li += 3;
            MPSR_REQUIRE(c1.kh == 1 && c1.cin == cur_c && c2.kh == 3 && c2.cin == c1.cout && c3.kh == 1 &&
                             c3.cin == c2.cout && (u == 0 || c3.cout == cur_c),
                         "trunk_fwd: records %d..%d do not form a bottleneck unit", li - 3, li - 1);
            float *dst = last ? out : (cur == ping ? pong : ping);
            // when conv1 runs on the persistent pointwise kernel and conv2 on the F(3x3,3x3) kernel, conv2's filter
            // transform rides on conv1's launch as a tail job (wino3_filter.h) instead of a launch of its own
            // (with a VALID filter cache there is nothing to transform; with one being filled the job writes the slice)
            float *uslice = cache.take(c2);
            const int c2i = (int)(&c2 - layers);
            const int fform = mpsr::winograd3_filter_form(B, d.PH, d.PW, c2.cin, c2.cout, c2.dilation);
            const bool cached_u = uslice && cache.valid && (!cache.tags || cache.tags[c2i] == fform);
            if (!cached_u &&
                mpsr::conv2d_takes_pointwise((long long)B * d.PH * d.PW, c1.cin, c1.cout, 1, 1, 0) &&
                mpsr::conv2d_takes_winograd3(B, d.PH, d.PW, c2.cin, c2.cout, c2.kh, c2.kw, c2.dilation, 0, sk, skn)) {
                mpsr::g_filter_tail_job.w = blob + c2.w_off;
                mpsr::g_filter_tail_job.u = uslice ? uslice : sk;
                mpsr::g_filter_tail_job.N = c2.cout;
                mpsr::g_filter_tail_job.C = c2.cin;
                mpsr::g_filter_tail_job.form = fform == mpsr::FILTER_FORM_WINO3Z ? 1 : 0;
            }
            if ((rc = run_layer(blob, c1, cur, B, d.PH, d.PW, nullptr, t1, 0, sk, skn, s))) return rc;
            mpsr::g_filter_tail_job = mpsr::FilterTailJob();  // (not taken: conv2 transforms its filters itself)
            cache.offer(blob + c2.w_off, uslice, c2, c2i);
            rc = run_layer(blob, c2, t1, B, d.PH, d.PW, nullptr, t2, 0, sk, skn, s);
            mpsr::g_filter_cache_slot = mpsr::FilterCacheSlot();
            if (rc) return rc;
            if ((rc = run_layer(blob, c3, t2, B, d.PH, d.PW, residual, dst, 0, sk, skn, s))) return rc;
            cur = dst;
            cur_c = c3.cout;
        }
    }
    return MPSR_OK;
}

// ------------------------------------------------------------------------------------------------ decoder

namespace {
// How mpsr_squash_decoder_fwd runs its layers -- one copy of the rule, mpsr_squash_decoder_plan reports it too.
//   c8:     the internal tensors are channel-blocked ([C/8][H][W][8]): every 3x3 layer runs on a kernel that reads it
//   up1/2:  conv2_1 / conv3_1 (each reads an upsampled map) as tap GEMM on the source map + gather (upconv.hip)
//   xyz_c8: the xyz head reads the last decoder layer's output channel-blocked
struct DecoderChoice {
    bool c8, up1, up2, xyz_c8;
};
DecoderChoice decoder_choice(int B, int fh, int fw, int mh, int mw, const mpsr_layer *L, const float *blob, bool feat_map,
                             bool xyz_map, const float *sk, size_t skn)
{
    const int hh = mh / 2, hw = mw / 2;
    const int csq = L[1].cout, c2 = L[2].cout, c3 = L[4].cout;
    DecoderChoice ch;
    // (the xyz head's own predicate: only its taps-in-N MFMA kernel reads a channel-blocked map)
    ch.xyz_c8 = !feat_map && xyz_map && L[6].kh == 3 && L[6].kw == 3 && L[6].dilation == 1 &&
                mpsr::conv3x3_narrow_takes_mfma(nullptr, B, mh, mw, L[6].cin, L[6].cout, blob ? blob + L[6].w_off : nullptr);
    const bool k3 = L[2].kh == 3 && L[3].kh == 3 && L[4].kh == 3 && L[5].kh == 3 && L[2].kw == 3 && L[3].kw == 3 &&
                    L[4].kw == 3 && L[5].kw == 3 && L[2].dilation == 1 && L[3].dilation == 1 && L[4].dilation == 1 &&
                    L[5].dilation == 1;
    // conv2_1 / conv3_1 read an upsampled map: tap GEMM on the source map + gather (fp32 arithmetic only)
    const bool upok = g_decoder_upconv.load() != 0 && k3 && mpsr_get_conv_math() == MPSR_MATH_FP32;
    ch.up1 = upok && mpsr::upconv_applies(B, fh, fw, csq, hh, hw, c2, 1) && skn >= mpsr::upconv_weight_floats(csq, c2);
    ch.up2 = upok && mpsr::upconv_applies(B, hh, hw, c2, mh, mw, c3, 1) && skn >= mpsr::upconv_weight_floats(c2, c3);
    ch.c8 = g_decoder_c8.load() != 0 && csq % 8 == 0 && c2 % 8 == 0 && c3 % 8 == 0 && k3 &&
            (ch.up1 || mpsr::conv2d_takes_winograd4(B, hh, hw, L[2].cin, L[2].cout, sk, skn)) &&
            mpsr::conv2d_takes_winograd4(B, hh, hw, L[3].cin, L[3].cout, sk, skn) &&
            (ch.up2 || mpsr::conv2d_takes_winograd4(B, mh, mw, L[4].cin, L[4].cout, sk, skn)) &&
            mpsr::conv2d_takes_winograd4(B, mh, mw, L[5].cin, L[5].cout, sk, skn);
    if (!ch.c8) ch.xyz_c8 = false;  // (the NHWC chain: conv2d per layer; the upsampled convolutions keep their tap GEMM)
    return ch;
}
}  // namespace

// Which kernel serves each of the seven decoder layers and the multiply-add FLOPs it issues (kinds as mpsr_conv2d_plan;
// 7 = tap GEMM on the source map + gather): what mpsr_squash_decoder_fwd does for this shape with xyz_map and without
// feat_map.  For throughput accounting (bench.py), not needed to run anything.
extern "C" int mpsr_squash_decoder_plan(int B, int fh, int fw, int mh, int mw, const mpsr_layer *L, int n_layers,
                                        int *kinds, double *executed_flops)
{
    MPSR_REQUIRE(B > 0 && fh >= 2 && fw >= 2 && mh >= 2 && mw >= 2 && L && n_layers == MPSR_DECODER_LAYERS && kinds &&
                     executed_flops,
                 "squash_decoder_plan: bad arguments");
    const int hh = mh / 2, hw = mw / 2;
    const size_t Mm = (size_t)B * mh * mw;
    int cwide = L[1].cout;
    for (int i = 2; i < n_layers; ++i) cwide = L[i].cout > cwide ? L[i].cout : cwide;
    const size_t skn = mpsr::conv_scratch_floats((long long)Mm, cwide > 512 ? cwide : 512);
    const DecoderChoice ch = decoder_choice(B, fh, fw, mh, mw, L, nullptr, false, true, reinterpret_cast<const float *>(256), skn);
    const int hs[7] = {fh, fh, hh, hh, mh, mh, mh}, ws[7] = {fw, fw, hw, hw, mw, mw, mw};
    for (int i = 0; i < 7; ++i) {
        int rc;
        if ((i == 2 && ch.up1) || (i == 4 && ch.up2)) {
            const int sh = i == 2 ? fh : hh, sw = i == 2 ? fw : hw;
            kinds[i] = 7;
            executed_flops[i] = 2.0 * (double)B * sh * sw * (double)L[i].cin * 9.0 * (double)L[i].cout;
        } else if (ch.c8 && (i == 2 || i == 3 || i == 4 || i == 5)) {
            kinds[i] = 3;
            executed_flops[i] = 2.0 * (double)B * (hs[i] / 4) * (ws[i] / 4) * 36.0 * (double)L[i].cin * (double)L[i].cout;
        } else if ((rc = mpsr_conv2d_plan(B, hs[i], ws[i], L[i].cin, L[i].cout, L[i].kh, L[i].kw, L[i].dilation, &kinds[i],
                                          &executed_flops[i])))
            return rc;
    }
    return MPSR_OK;
}

extern "C" size_t mpsr_decoder_workspace_bytes(int B, int fh, int fw, int mh, int mw)
{
    if (B <= 0) return 0;
    const size_t Mf = (size_t)B * fh * fw, Mh = (size_t)B * (mh / 2) * (mw / 2), Mm = (size_t)B * mh * mw;
    // (+ with the opt-in position-split Winograd kernel enabled, the partial outputs it parks: the larger map size)
    const size_t parth = mpsr::winograd4_split_floats(B, mh / 2, mw / 2, 256), partm = mpsr::winograd4_split_floats(B, mh, mw, 128);
    const size_t part = mpsr::g_wino4_split.load() != 0 ? fbytes((parth > partm ? parth : partm) + 64) : 0;
    // (the two upsampled maps' slots also hold the tap GEMMs' outputs, 9 x 256 / 9 x 128 columns at the source size)
    const size_t r1 = Mh * 512 > Mf * 9 * 256 ? Mh * 512 : Mf * 9 * 256, r2 = Mm * 256 > Mh * 9 * 128 ? Mm * 256 : Mh * 9 * 128;
    return 2 * fbytes(Mf * 512) + fbytes(r1) + 2 * fbytes(Mh * 256) + fbytes(r2) + 2 * fbytes(Mm * 128) +
           fbytes(mpsr::conv_scratch_floats((long long)Mm, 512)) + part;
}

extern "C" int mpsr_squash_decoder_fwd(const float *crop_feat, const float *full_feat, int B, int fh, int fw, int mh,
                                       int mw, const float *blob, const mpsr_layer *L, int n_layers,
                                       float *feat_box3d, float *feat_map, float *xyz_map, void *workspace,
                                       size_t workspace_bytes, mpsr_stream_t stream)
{
    return mpsr_squash_decoder_fwd_ex(crop_feat, full_feat, B, fh, fw, mh, mw, blob, L, n_layers, feat_box3d, feat_map,
                                      xyz_map, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int mpsr_squash_decoder_fwd_ex(const float *crop_feat, const float *full_feat, int B, int fh, int fw, int mh,
                                          int mw, const float *blob, const mpsr_layer *L, int n_layers,
                                          float *feat_box3d, float *feat_map, float *xyz_map, void *workspace,
                                          size_t workspace_bytes, const mpsr_net_opts *opts, mpsr_stream_t stream)
{
    MPSR_REQUIRE(B >= 0 && fh >= 2 && fw >= 2 && mh >= 2 && mw >= 2 && mh % 2 == 0 && mw % 2 == 0,
                 "squash_decoder_fwd: bad shape");
    if (opts) {
        MPSR_REQUIRE(opts->math >= 0 && opts->math <= MPSR_CALL_MATH_BF16X3, "%s: unknown opts.math %d", "squash_decoder_fwd", opts->math);
        MPSR_REQUIRE(opts->winograd_policy >= 0 && opts->winograd_policy <= MPSR_CALL_WINOGRAD_ACCURATE,
                     "%s: unknown opts.winograd_policy %d", "squash_decoder_fwd", opts->winograd_policy);
    }
    mpsr::CallOptsGuard call_opts(opts ? opts->math : 0, opts ? opts->winograd_policy : 0);
    MPSR_REQUIRE(n_layers == MPSR_DECODER_LAYERS, "squash_decoder_fwd: expected %d layer records, got %d",
                 MPSR_DECODER_LAYERS, n_layers);
    if (B == 0) return MPSR_OK;
    MPSR_REQUIRE(crop_feat && full_feat && blob && L && feat_box3d && workspace,
                 "squash_decoder_fwd: null pointer");
    hipStream_t s = mpsr::as_stream(stream);
    const int hh = mh / 2, hw = mw / 2;
    const size_t Mf = (size_t)B * fh * fw, Mh = (size_t)B * hh * hw, Mm = (size_t)B * mh * mw;
    const int csq = L[1].cout, c2 = L[2].cout, c3 = L[4].cout;
    MPSR_REQUIRE(L[0].cout == csq && L[2].cin == csq && L[3].cin == c2 && L[4].cin == c2 && L[5].cin == c3 &&
                     L[6].cin == c3,
                 "squash_decoder_fwd: layer records are not squash_a, squash_b, conv2_1, conv2_2, conv3_1, conv3_2, xyz");
    Arena ar(workspace, workspace_bytes);
    float *part = ar.floats(Mf * csq), *sq = ar.floats(Mf * csq);
    // r1 / r2: the upsampled maps, or -- same slots -- the tap GEMMs' outputs z when the upsampled convolutions run on
    // upconv.hip (9 x cout columns at the SOURCE resolution)
    const size_t r1n = Mh * csq > Mf * 9 * (size_t)c2 ? Mh * csq : Mf * 9 * (size_t)c2;
    const size_t r2n = Mm * c2 > Mh * 9 * (size_t)c3 ? Mm * c2 : Mh * 9 * (size_t)c3;
    float *r1 = ar.floats(r1n), *a = ar.floats(Mh * c2), *b = ar.floats(Mh * c2);
    float *r2 = ar.floats(r2n), *c = ar.floats(Mm * c3);
    float *fm = feat_map ? feat_map : ar.floats(Mm * c3);
    int cwide = csq;
    for (int i = 2; i < n_layers; ++i) cwide = L[i].cout > cwide ? L[i].cout : cwide;
    const size_t skn = mpsr::conv_scratch_floats((long long)Mm, cwide > 512 ? cwide : 512);
    float *sk = ar.floats(skn);
    // optional: partial outputs of the position-split Winograd kernel (absent in a workspace sized by an older
    // mpsr_decoder_workspace_bytes: the 64-channel-block kernel then runs)
    size_t partn = 0;
    float *partb = nullptr;
    if (ar.ok && mpsr::g_wino4_split.load() != 0 && c2 % 128 == 0 && c3 % 128 == 0 && hh % 4 == 0 && hw % 4 == 0 && mh % 4 == 0 && mw % 4 == 0) {
        const size_t a1 = mpsr::winograd4_split_floats(B, hh, hw, c2), a2 = mpsr::winograd4_split_floats(B, mh, mw, c3);
        partn = (a1 > a2 ? a1 : a2) + 64;
        partb = ar.floats(partn);
        if (!ar.ok) {
            ar.ok = true;
            partb = nullptr;
            partn = 0;
        }
    }
    if (!ar.ok)
        return mpsr::fail(MPSR_ERR_WORKSPACE, "squash_decoder_fwd: workspace %zu bytes too small", workspace_bytes);
    int rc;
    // 1x1 over concat([crop, full]) == two GEMMs over the halves of K; the second adds the first as its residual
    if ((rc = run_layer(blob, L[0], crop_feat, B, fh, fw, nullptr, part, 0, sk, skn, s))) return rc;
    if ((rc = run_layer(blob, L[1], full_feat, B, fh, fw, part, sq, 0, sk, skn, s))) return rc;
    if ((rc = mpsr_max_pool(sq, B, fh, fw, csq, 2, 2, 0, feat_box3d, stream))) return rc;
    // features_for_box_3d is complete: the caller's FC heads may start on another stream
    if (opts && opts->ready_event) MPSR_CHECK_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(opts->ready_event), s));
    struct SlotGuard {
        ~SlotGuard() { mpsr::g_filter_cache_slot = mpsr::FilterCacheSlot(); }
    } slot_guard;
    FilterCache cache(opts);
    // Channel-blocked internal tensors ([C/8][H][W][8], "C8") when all four 3x3 layers run on the F(4x4,3x3) kernel: a
    // K step of that kernel then reads whole 128-byte lines instead of 32 bytes of every pixel's line (NHWC re-fetched
    // the lines from beyond L2: 1.87x the layers' own bytes, profiles/r03_*), and the xyz head's A loads become 1 KiB
    // contiguous.  Layouts: r1, a, r2, c are C8; b (the input of the second resize) and a requested feat_map are NHWC.
    // Same kernels, same arithmetic order: bit-identical to the NHWC chain (tests/test_net_gpu.py).
    const DecoderChoice ch = decoder_choice(B, fh, fw, mh, mw, L, blob, feat_map != nullptr, xyz_map != nullptr, sk, skn);
    const bool xyz_c8 = ch.xyz_c8, up1 = ch.up1, up2 = ch.up2, c8 = ch.c8;
    if (c8) {
        auto wino = [&](const mpsr_layer &Lr, const float *x, int H, int W, float *y, int in_c8, int out_c8) {
            cache.offer(blob + Lr.w_off, cache.take(Lr), Lr, (int)(&Lr - L));
            return mpsr::conv3x3_winograd4(x, B, H, W, Lr.cin, blob + Lr.w_off, Lr.b_off >= 0 ? blob + Lr.b_off : nullptr,
                                           Lr.relu, y, Lr.cout, sk, skn, s, in_c8, out_c8, partb, partn);
        };
        // upsampling + 3x3 convolution in one: x (NHWC, source size) -> y (channel-blocked, output size); z = r1 / r2
        auto upconv = [&](const mpsr_layer &Lr, const float *x, int h, int w, int H, int W, float *y, float *z, size_t zn) {
            cache.offer(blob + Lr.w_off, cache.take(Lr), Lr, (int)(&Lr - L));
            return mpsr::conv3x3_upsampled(x, B, h, w, Lr.cin, H, W, 1, blob + Lr.w_off,
                                           Lr.b_off >= 0 ? blob + Lr.b_off : nullptr, Lr.relu, y, Lr.cout, 1, z, zn, sk,
                                           skn, s);
        };
        if (up1) {
            if ((rc = upconv(L[2], sq, fh, fw, hh, hw, a, r1, r1n))) return rc;
        } else {
            if ((rc = mpsr::resize_bilinear_c8(sq, B, fh, fw, csq, hh, hw, 1, r1, s))) return rc;
            if ((rc = wino(L[2], r1, hh, hw, a, 1, 1))) return rc;
        }
        if ((rc = wino(L[3], a, hh, hw, b, 1, 0))) return rc;
        if (up2) {
            if ((rc = upconv(L[4], b, hh, hw, mh, mw, c, r2, r2n))) return rc;
        } else {
            if ((rc = mpsr::resize_bilinear_c8(b, B, hh, hw, c2, mh, mw, 1, r2, s))) return rc;
            if ((rc = wino(L[4], r2, mh, mw, c, 1, 1))) return rc;
        }
        if ((rc = wino(L[5], c, mh, mw, fm, 1, xyz_c8 ? 1 : 0))) return rc;
        if (xyz_map) {
            if (xyz_c8)
                rc = mpsr::conv3x3_narrow(fm, B, mh, mw, L[6].cin, blob + L[6].w_off,
                                          L[6].b_off >= 0 ? blob + L[6].b_off : nullptr, L[6].relu, xyz_map, L[6].cout, s, 1);
            else
                rc = run_layer(blob, L[6], fm, B, mh, mw, nullptr, xyz_map, 1, nullptr, 0, s);
            if (rc) return rc;
        }
        return MPSR_OK;
    }
    // NHWC chain (small batches, maps that do not divide into 4x4 blocks, MPSR_WINOGRAD_OFF, the bf16x3 mode): conv2d
    // per layer; the two upsampled convolutions keep their tap GEMM + gather (NHWC output) where it applies
    auto cached = [&](const mpsr_layer &Lr, const float *x, int H, int W, float *y) {
        cache.offer(blob + Lr.w_off, cache.take(Lr), Lr, (int)(&Lr - L));
        return run_layer(blob, Lr, x, B, H, W, nullptr, y, 0, sk, skn, s);
    };
    auto upconv_nhwc = [&](const mpsr_layer &Lr, const float *x, int h, int w, int H, int W, float *y, float *z, size_t zn) {
        cache.offer(blob + Lr.w_off, cache.take(Lr), Lr, (int)(&Lr - L));
        return mpsr::conv3x3_upsampled(x, B, h, w, Lr.cin, H, W, 1, blob + Lr.w_off,
                                       Lr.b_off >= 0 ? blob + Lr.b_off : nullptr, Lr.relu, y, Lr.cout, 0, z, zn, sk, skn, s);
    };
    if (up1) {
        if ((rc = upconv_nhwc(L[2], sq, fh, fw, hh, hw, a, r1, r1n))) return rc;
    } else {
        if ((rc = mpsr_resize_bilinear(sq, B, fh, fw, csq, hh, hw, 1, r1, stream))) return rc;
        if ((rc = cached(L[2], r1, hh, hw, a))) return rc;
    }
    if ((rc = cached(L[3], a, hh, hw, b))) return rc;
    if (up2) {
        if ((rc = upconv_nhwc(L[4], b, hh, hw, mh, mw, c, r2, r2n))) return rc;
    } else {
        if ((rc = mpsr_resize_bilinear(b, B, hh, hw, c2, mh, mw, 1, r2, stream))) return rc;
        if ((rc = cached(L[4], r2, mh, mw, c))) return rc;
    }
    if ((rc = cached(L[5], c, mh, mw, fm))) return rc;
    if (xyz_map && (rc = run_layer(blob, L[6], fm, B, mh, mw, nullptr, xyz_map, 1, nullptr, 0, s))) return rc;
    return MPSR_OK;
}

// ------------------------------------------------------------------------------------------------ heads

namespace {

struct HeadIn {
    const float *boxes, *cam_p, *view, *mean_lwh, *z_off;
    const int *cls;
    const int *cam_idx;  // per box: which of the cam_p matrices (boxes of several images in one call), or nullptr: the first
    int n_cams;          // cam_idx[b] outside [0, n_cams) reads kBadCam: that box's outputs are NaN, nothing is read out of bounds
    mpsr_head_consts k;
};
__device__ const float kBadCam[12] = {__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""),
                                      __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""),
                                      __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
__device__ __forceinline__ const float *cam_of(const HeadIn &in, int b)
{
    if (!in.cam_idx) return in.cam_p;
    const int i = in.cam_idx[b];  // (a caller's index: guarded like crop_and_resize's box_ind, image_ops.hip)
    return (i >= 0 && i < in.n_cams) ? in.cam_p + 12 * i : kBadCam;
}

// scalar features shared by both concat rows (monopsr_output_builder.py:147-158,228-240):
// index 0..3 box coords in film coordinates / half image size, 4 box height / image height, 5 view angle,
// 6..6+nc-1 class one-hot
__device__ __forceinline__ float common_feature(const HeadIn &in, int b, int j)
{
    const float *bx = in.boxes + 4 * b;
    const float *cam = cam_of(in, b);
    const float cu = cam[2], cv = cam[6];
    if (j < 4) {
        const float centre = (j & 1) ? cu : cv;
        const float half = ((j & 1) ? in.k.image_w : in.k.image_h) / 2.0f;
        return (bx[j] - centre) / half;
    }
    if (j == 4) return (bx[2] - bx[0]) / in.k.image_h;
    if (j == 5) return in.view[b];
    return (in.cls[b] == j - 6) ? 1.0f : 0.0f;
}

__constant__ float kCamNorm[12] = {1000.f, 1.f, 1000.f, 100.f, 1.f, 1000.f, 1000.f, 1.f, 1.f, 1.f, 1.f, 1.f};

// row b of the proposal FC input: [img_fc(1024) | common | cam_p / norm (12) | zero pad]
__global__ __launch_bounds__(256) void head_concat_prop_kernel(HeadIn in, const float *__restrict__ imgfc,
                                                               int imgfc_stride, int nfc, int kpad, int B,
                                                               float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * kpad) return;
    const int b = i / kpad, col = i - b * kpad;
    const int ncommon = 6 + in.k.num_classes;
    float v = 0.f;
    if (col < nfc) v = imgfc[(size_t)b * imgfc_stride + col];
    else if (col < nfc + ncommon) v = common_feature(in, b, col - nfc);
    else if (col < nfc + ncommon + 12) v = cam_of(in, b)[col - nfc - ncommon] / kCamNorm[col - nfc - ncommon];
    out[i] = v;
}

// per box: lwh / alpha outputs, centroid proposals (get_prop_cen_z :407-431, tf_est_y_from_box_2d_and_depth
// instance_utils.py:907-953).  pout row = [lwh_offs(3) | alpha_bins(nb) | alpha_regs(nb)].  prop = [y, z] per box.
__global__ __launch_bounds__(256) void head_prop_outputs_kernel(HeadIn in, const float *__restrict__ pout, int B,
                                                                mpsr_head_outputs o, float *__restrict__ prop)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int nb = in.k.num_alpha_bins, stride = 3 + 2 * nb;
    const float *p = pout + (size_t)b * stride;
    float lwh[3];
    for (int j = 0; j < 3; ++j) {
        lwh[j] = in.mean_lwh[3 * b + j] + p[j];
        if (o.lwh) o.lwh[3 * b + j] = lwh[j];
        if (o.lwh_offs) o.lwh_offs[3 * b + j] = p[j];
    }
    for (int j = 0; j < nb; ++j) {
        if (o.alpha_bins) o.alpha_bins[(size_t)b * nb + j] = p[3 + j];
        if (o.alpha_regs) o.alpha_regs[(size_t)b * nb + j] = p[3 + nb + j];
    }
    const float *bx = in.boxes + 4 * b;
    const float focal = cam_of(in, b)[0], cv = cam_of(in, b)[6];
    const float z = focal * lwh[2] / (bx[2] - bx[0]) + in.z_off[b];
    const float centre_v = (bx[2] + bx[0]) / 2.0f - cv;
    const float y = centre_v * (z / focal) - in.k.cen_y_class_offset;
    if (o.prop_cen_z) o.prop_cen_z[b] = z;
    prop[2 * b] = y;
    prop[2 * b + 1] = z;
}

// row b of the regression FC input: [img_fc_r(1024) | common | lwh_offs(3) | bins | regs | y/norm | z/max_depth | pad]
__global__ __launch_bounds__(256) void head_concat_reg_kernel(HeadIn in, const float *__restrict__ imgfc,
                                                              int imgfc_stride, int nfc, int kpad, int B,
                                                              const float *__restrict__ pout,
                                                              const float *__restrict__ prop, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * kpad) return;
    const int b = i / kpad, col = i - b * kpad;
    const int ncommon = 6 + in.k.num_classes, nout = 3 + 2 * in.k.num_alpha_bins;
    float v = 0.f;
    if (col < nfc) v = imgfc[(size_t)b * imgfc_stride + nfc + col];
    else if (col < nfc + ncommon) v = common_feature(in, b, col - nfc);
    else if (col < nfc + ncommon + nout) v = pout[(size_t)b * nout + (col - nfc - ncommon)];
    else if (col == nfc + ncommon + nout) v = prop[2 * b] / in.k.cen_y_norm;
    else if (col == nfc + ncommon + nout + 1) v = prop[2 * b + 1] / in.k.max_depth;
    out[i] = v;
}

// per box: centroid = proposal + regressed offsets; x from the viewing angle (add_cen_x_output :551-571)
__global__ __launch_bounds__(256) void head_final_kernel(HeadIn in, const float *__restrict__ rout,
                                                         const float *__restrict__ prop, int B, mpsr_head_outputs o)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float yo = rout[2 * b], zo = rout[2 * b + 1];
    const float y = prop[2 * b] + yo, z = prop[2 * b + 1] + zo;
    const float x = z * tanf(in.view[b]) + (-cam_of(in, b)[3] / cam_of(in, b)[0]);
    if (o.cen_y) o.cen_y[b] = y;
    if (o.cen_y_offs) o.cen_y_offs[b] = yo;
    if (o.cen_z) o.cen_z[b] = z;
    if (o.cen_z_offs) o.cen_z_offs[b] = zo;
    if (o.cen_x) o.cen_x[b] = x;
    if (o.centroids) {
        o.centroids[3 * b] = x;
        o.centroids[3 * b + 1] = y;
        o.centroids[3 * b + 2] = z;
    }
}

}  // namespace

extern "C" size_t mpsr_heads_workspace_bytes(int B, int feat_elems)
{
    if (B <= 0 || feat_elems <= 0) return 0;
    const size_t b = (size_t)B;
    // img_fc (2 x 1024 wide) + the scheduling scratch of its launch (stream-K partial tiles), two concat rows, four
    // hidden activations, small outputs
    return fbytes(b * 2048) + fbytes(mpsr::conv_scratch_floats(B, 2048)) + 2 * fbytes(b * 1152) + 4 * fbytes(b * 1024) +
           fbytes(b * 64) + fbytes(b * 2) + fbytes(b * 2);
}

extern "C" int mpsr_heads_fwd(const float *feat_box3d, int B, int feat_elems, const float *boxes_2d,
                              const float *cam_p, const float *view_angs, const int *class_idx, const float *mean_lwh,
                              const float *cen_z_offset, const mpsr_head_consts *consts, const float *blob,
                              const mpsr_layer *L, int n_layers, const mpsr_head_outputs *outs, void *workspace,
                              size_t workspace_bytes, mpsr_stream_t stream)
{
    return mpsr_heads_fwd_cams(feat_box3d, B, feat_elems, boxes_2d, cam_p, 1, nullptr, view_angs, class_idx, mean_lwh,
                               cen_z_offset, consts, blob, L, n_layers, outs, workspace, workspace_bytes, stream);
}

// The boxes of SEVERAL images in one call: cam_p (n_cams,12), cam_index (B) in [0, n_cams) selects each box's projection
// matrix (nullptr: all boxes use the first; an index outside the range makes that box's outputs NaN -- never an
// out-of-bounds read).  The reference's step is one image (monopsr_model.py:95, one pl_cam_p); a
// batch of images is N such steps whose FC weights (150 MB) are then read once instead of N times.
extern "C" int mpsr_heads_fwd_cams(const float *feat_box3d, int B, int feat_elems, const float *boxes_2d,
                                   const float *cam_p, int n_cams, const int *cam_index, const float *view_angs,
                                   const int *class_idx, const float *mean_lwh, const float *cen_z_offset,
                                   const mpsr_head_consts *consts, const float *blob, const mpsr_layer *L, int n_layers,
                                   const mpsr_head_outputs *outs, void *workspace, size_t workspace_bytes,
                                   mpsr_stream_t stream)
{
    MPSR_REQUIRE(n_cams >= 1, "heads_fwd: n_cams = %d", n_cams);
    MPSR_REQUIRE(B >= 0 && feat_elems > 0 && feat_elems % 4 == 0, "heads_fwd: bad shape (B=%d feat=%d)", B, feat_elems);
    MPSR_REQUIRE(n_layers == MPSR_HEAD_LAYERS, "heads_fwd: expected %d layer records, got %d", MPSR_HEAD_LAYERS,
                 n_layers);
    if (B == 0) return MPSR_OK;
    MPSR_REQUIRE(feat_box3d && boxes_2d && cam_p && view_angs && class_idx && mean_lwh && cen_z_offset && consts &&
                     blob && L && outs && workspace,
                 "heads_fwd: null pointer");
    const int nfc = L[0].cout / 2;  // width of each img_fc
    const int nb = consts->num_alpha_bins, nc = consts->num_classes;
    const int prop_in = nfc + 6 + nc + 12, reg_in = nfc + 6 + nc + 3 + 2 * nb + 2;
    MPSR_REQUIRE(L[0].cin == feat_elems && L[1].cin >= prop_in && L[1].cin % 4 == 0 && L[3].cout == 3 + 2 * nb &&
                     L[4].cin >= reg_in && L[4].cin % 4 == 0 && L[6].cout == 2 && L[1].cin <= 1152 && L[4].cin <= 1152 &&
                     L[1].cout <= 1024 && L[2].cout <= 1024 && L[4].cout <= 1024 && L[5].cout <= 1024 && nfc <= 1024 &&
                     3 + 2 * nb <= 64,
                 "heads_fwd: layer records do not match the head graph");
    hipStream_t s = mpsr::as_stream(stream);
    Arena ar(workspace, workspace_bytes);
    const size_t b = (size_t)B;
    float *imgfc = ar.floats(b * L[0].cout);
    const size_t skn = mpsr::conv_scratch_floats(B, L[0].cout > 2048 ? L[0].cout : 2048);
    float *skws = ar.floats(skn);
    float *cat_p = ar.floats(b * L[1].cin), *cat_r = ar.floats(b * L[4].cin);
    float *h1 = ar.floats(b * L[1].cout), *h2 = ar.floats(b * L[2].cout);
    float *g1 = ar.floats(b * L[4].cout), *g2 = ar.floats(b * L[5].cout);
    float *pout = ar.floats(b * L[3].cout), *rout = ar.floats(b * 2), *prop = ar.floats(b * 2);
    if (!ar.ok) return mpsr::fail(MPSR_ERR_WORKSPACE, "heads_fwd: workspace %zu bytes too small", workspace_bytes);

    HeadIn in;
    in.boxes = boxes_2d; in.cam_p = cam_p; in.view = view_angs; in.mean_lwh = mean_lwh; in.z_off = cen_z_offset;
    in.cls = class_idx; in.cam_idx = cam_index; in.n_cams = n_cams; in.k = *consts;
    int rc;
    // both img_fc layers share the flattened features: one GEMM, N = 2 x 1024, split along K = 18432
    if ((rc = run_layer(blob, L[0], feat_box3d, B, 1, 1, nullptr, imgfc, 0, skws, skn, s))) return rc;
    {
        const int total = B * L[1].cin;
        hipLaunchKernelGGL(head_concat_prop_kernel, dim3(mpsr::ceil_div(total, 256)), dim3(256), 0, s, in, imgfc,
                           L[0].cout, nfc, L[1].cin, B, cat_p);
        MPSR_CHECK_LAUNCH("head_concat_prop_kernel");
    }
    if ((rc = run_layer(blob, L[1], cat_p, B, 1, 1, nullptr, h1, 1, nullptr, 0, s))) return rc;
    if ((rc = run_layer(blob, L[2], h1, B, 1, 1, nullptr, h2, 1, nullptr, 0, s))) return rc;
    if ((rc = run_layer(blob, L[3], h2, B, 1, 1, nullptr, pout, 1, nullptr, 0, s))) return rc;
    hipLaunchKernelGGL(head_prop_outputs_kernel, dim3(mpsr::ceil_div(B, 256)), dim3(256), 0, s, in, pout, B, *outs, prop);
    MPSR_CHECK_LAUNCH("head_prop_outputs_kernel");
    {
        const int total = B * L[4].cin;
        hipLaunchKernelGGL(head_concat_reg_kernel, dim3(mpsr::ceil_div(total, 256)), dim3(256), 0, s, in, imgfc,
                           L[0].cout, nfc, L[4].cin, B, pout, prop, cat_r);
        MPSR_CHECK_LAUNCH("head_concat_reg_kernel");
    }
    if ((rc = run_layer(blob, L[4], cat_r, B, 1, 1, nullptr, g1, 1, nullptr, 0, s))) return rc;
    if ((rc = run_layer(blob, L[5], g1, B, 1, 1, nullptr, g2, 1, nullptr, 0, s))) return rc;
    if ((rc = run_layer(blob, L[6], g2, B, 1, 1, nullptr, rout, 1, nullptr, 0, s))) return rc;
    hipLaunchKernelGGL(head_final_kernel, dim3(mpsr::ceil_div(B, 256)), dim3(256), 0, s, in, rout, prop, B, *outs);
    MPSR_CHECK_LAUNCH("head_final_kernel");
    return MPSR_OK;
}
