// plan.hip — whole-network execution plan for UNet3D: workspace layout + forward / backward kernel sequences.
// One C call launches the entire forward (or a range of backward segments) on the caller's stream; nothing
// here touches Python, allocates, or synchronises, so a step is hipGraph-capturable end to end.
//
// Reference: models/unet.py:34-90 (UNet3D), models/unet_dann.py:65-98 (GAP branch).
// Data layout in HBM (all inside the caller-owned workspace, channels-last, dtype T):
//   per DoubleConv half : y (raw conv output, kept for BN backward), stat[4][C]
//   per block           : z1 (activated first half)
//   per level l         : cat[l]  [N,V_l,2C_l]  = [ encoder output | upconv output ]   (torch.cat is free)
//                         pool[l] [N,V_l/8,C_l]
//   gradients           : gz[l], gcat[l], gp[l] per level + two max-size scratch tensors
#include "../../include/mi3d.h"
#include <stdlib.h>

#include "ops.h"

namespace {

constexpr int MAXL = MI3D_MAX_LEVELS;

struct HalfP {
    int Cin, Cout;
    size_t y, stat, wpf, wpd;     // byte offsets
    int pidx, bidx;
    int64_t drop_off;
    bool mfma;                    // bf16 implicit-GEMM path (conv3_mfma.hip) vs direct fp32-FMA path
    // Deferred weight gradient (round 4): with an aux stream the layer's dy gets its OWN buffer (dyk) that stays alive, the
    // data-gradient chain runs the input-gradient conv alone, and the weight gradient is launched later on the aux stream.
    //   1 = decoder layer at a 16-wide-tile level (levels 0-1 at 96^3): runs under the latency-bound deep-level chain
    //   2 = any layer at a deep level (8-wide tiles): runs under the bandwidth-bound encoder backward of levels 1-0
    //   0 = encoder layers at the 16-wide levels: nothing left to hide under, they keep the fused launch
    int defer;
    size_t dyk;
};
struct BlockP {
    int level;
    HalfP h[2];
    size_t z1;
};
struct Plan {
    mi3d_unet_desc d;
    int L, dt;
    size_t esz;
    int C[MAXL + 1];
    Geo geo[MAXL + 1];
    BlockP blk[2 * MAXL + 1];
    int nblk;
    size_t cat[MAXL], pool[MAXL], zb, zd[MAXL], upw[MAXL], xcl;
    bool up_mfma[MAXL];
    // planar[l]: the level's concat buffers cat[l] / gcat[l] hold the skip half and the up half as two [M][C] planes
    // instead of one interleaved [M][2C] tensor.  With C = 16 the interleaved halves are 32 B pieces of 64 B rows and
    // every kernel that touches ONE half (bn apply, max-pool fwd/bwd, upconv fwd/bwd) wastes half of each line.
    bool planar[MAXL];
    // resize[l]: the transposed conv's output (2x the level below) is smaller than the skip at level l (odd side somewhere
    // above) -> it goes to uptmp and is nearest-resized into the concat buffer (models/unet.py:81-83)
    bool resize[MAXL];
    size_t uptmp;
    Geo up_geo(int l) const { return Geo{geo[l + 1].N, 2 * geo[l + 1].D, 2 * geo[l + 1].H, 2 * geo[l + 1].W}; }
    int catcs(int l) const { return planar[l] ? C[l] : 2 * C[l]; }
    size_t half_off(int l) const { return (planar[l] ? (size_t)geo[l].M() * C[l] : (size_t)C[l]) * esz; }   // bytes to the up half
    Halves halves(int l) const {
        Halves h;
        if (planar[l]) { h.split = C[l] / 16; h.delta = geo[l].M() * C[l] - C[l]; }
        return h;
    }
    size_t gz[MAXL + 1], gcat[MAXL], gp[MAXL], sB, sB2, sC;
    size_t bnws, wgws, wgws2, wgws3, statpart, skws, tkcount;
    // (block, half) after whose BatchNorm backward the pending deferred weight gradients of group 1 / 2 go to the aux stream
    int flush_b[2], flush_h[2];
    size_t wgws_floats;
    size_t total;
    int up_pidx(int i) const { return 8 * (L + 1) + 2 * i; }
    int final_pidx() const { return 8 * (L + 1) + 2 * L + 8 * L; }
};

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

int build_plan(const mi3d_unet_desc* d, Plan& p) {
    MI3D_CHECK_ARG(d != nullptr, "null descriptor");
    MI3D_CHECK_ARG(d->n_levels >= 1 && d->n_levels <= MAXL, "n_levels=%d out of range", d->n_levels);
    MI3D_CHECK_ARG(d->dtype == MI3D_F32 || d->dtype == MI3D_BF16, "bad dtype %d", d->dtype);
    MI3D_CHECK_ARG(d->in_channels >= 1 && d->out_channels >= 1 && d->out_channels <= MI3D_MAX_CLASSES,
                   "unsupported channel counts in=%d out=%d (out <= %d)", d->in_channels, d->out_channels, MI3D_MAX_CLASSES);
    MI3D_CHECK_ARG(d->N >= 1 && d->D >= 1 && d->H >= 1 && d->W >= 1, "bad shape");
    p.d = *d;
    p.L = d->n_levels;
    p.dt = d->dtype;
    p.esz = d->dtype == MI3D_F32 ? 4 : 2;
    // sides not divisible by 2^L: MaxPool3d floors and models/unet.py:81-83 nearest-resizes the upsampled tensor to the
    // skip's shape before the concat (resize[l]); every level must keep at least one voxel per side
    MI3D_CHECK_ARG((d->D >> p.L) >= 1 && (d->H >> p.L) >= 1 && (d->W >> p.L) >= 1,
                   "volume %dx%dx%d too small for %d pooling levels", d->D, d->H, d->W, p.L);
    for (int l = 0; l < p.L; l++) {
        MI3D_CHECK_ARG(d->features[l] >= 1 && d->features[l] <= 256, "feature %d out of range", d->features[l]);
        if (l > 0) MI3D_CHECK_ARG(d->features[l] == 2 * d->features[l - 1], "features must double per level");
        p.C[l] = d->features[l];
    }
    p.C[p.L] = 2 * d->features[p.L - 1];
    MI3D_CHECK_ARG(p.C[p.L] <= 256, "bottleneck width %d > 256", p.C[p.L]);
    for (int l = 0; l <= p.L; l++) p.geo[l] = Geo{d->N, d->D >> l, d->H >> l, d->W >> l};
    size_t up_elems = 0;
    for (int l = 0; l < p.L; l++) {
        Geo u = p.up_geo(l);
        p.resize[l] = u.D != p.geo[l].D || u.H != p.geo[l].H || u.W != p.geo[l].W;
        if (p.resize[l] && (size_t)u.M() * d->features[l] > up_elems) up_elems = (size_t)u.M() * d->features[l];
    }

    for (int l = 0; l < p.L; l++)
        p.planar[l] = p.dt == MI3D_BF16 && p.C[l] % 16 == 0 && conv3_mfma_halves_ok(2 * p.C[l], p.C[l], p.geo[l]) &&
                      conv3_mfma_halves_ok(p.C[l], 2 * p.C[l], p.geo[l]) && !mi3d_routes().force_direct && !mi3d_routes().no_planar;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
    int64_t drop_off = 0;
    size_t wg_floats = 0, maxCM = 0, statpart_floats = 1, skws_floats = 1;
    int maxC = 1;
    p.nblk = 2 * p.L + 1;
    for (int b = 0; b < p.nblk; b++) {
        BlockP& B = p.blk[b];
        int cin, cout;
        if (b < p.L) { B.level = b; cin = b == 0 ? d->in_channels : p.C[b - 1]; cout = p.C[b]; }
        else if (b == p.L) { B.level = p.L; cin = p.C[p.L - 1]; cout = p.C[p.L]; }
        else { int i = b - p.L - 1; B.level = p.L - 1 - i; cin = 2 * p.C[B.level]; cout = p.C[B.level]; }
        Geo g = p.geo[B.level];
        int pbase = b <= p.L ? 8 * b : 8 * (p.L + 1) + 2 * p.L + 8 * (b - p.L - 1);
        for (int h = 0; h < 2; h++) {
            HalfP& H = B.h[h];
            H.Cin = h == 0 ? cin : cout;
            H.Cout = cout;
            H.y = take((size_t)g.M() * cout * p.esz);
            H.stat = take((size_t)4 * cout * sizeof(float));
            H.mfma = p.dt == MI3D_BF16 && conv3_mfma_supported(H.Cin, H.Cout, 16, 16) && !mi3d_routes().force_direct;
            if (H.mfma) {
                H.wpf = take(conv3_mfma_pack_elems(H.Cin, H.Cout) * 2);
                H.wpd = take(conv3_mfma_pack_elems(H.Cin, H.Cout) * 2);
                size_t sp = (size_t)conv3_mfma_stat_blocks(H.Cin, H.Cout, g) * 2 * cout;
                if (sp > statpart_floats) statpart_floats = sp;
                size_t sk = conv3_mfma_splitk_floats(H.Cin, H.Cout, g), sk2 = conv3_mfma_splitk_floats(H.Cout, H.Cin, g);
                if (sk > skws_floats) skws_floats = sk;
                if (sk2 > skws_floats) skws_floats = sk2;
            } else {
                H.wpf = take(conv3_direct_pack_floats(H.Cin, H.Cout) * sizeof(float));
                H.wpd = take(conv3_direct_pack_floats(H.Cout, H.Cin) * sizeof(float));
            }
            H.pidx = pbase + 4 * h;
            H.bidx = 6 * b + 3 * h;
            H.drop_off = drop_off;
            drop_off += (int64_t)d->N * cout;
            bool c1 = p.dt == MI3D_BF16 && H.Cin == 1 && H.Cout % 16 == 0 && !mi3d_routes().force_direct;
            if (c1) {
                size_t sp = (size_t)conv3_c1_fwd_stat_blocks(g) * 2 * cout;
                if (sp > statpart_floats) statpart_floats = sp;
            }
            size_t wf = (H.mfma || c1) ? conv3_mfma_wgrad_ws_floats(H.Cin, H.Cout, g) : conv3_direct_wgrad_ws_floats(H.Cin, H.Cout, g);
            if (wf > wg_floats) wg_floats = wf;
            // the dy buffers are part of the layout whatever the route says (a route changed between the workspace query and a
            // launch must not move anything)
            H.defer = !H.mfma ? 0 : !conv3_mfma_big_geo(g) ? 2 : (b > p.L ? 1 : 0);
            H.dyk = H.defer ? take((size_t)g.M() * cout * p.esz) : 0;
        }
        B.z1 = take((size_t)g.M() * cout * p.esz);
        if ((size_t)g.M() * cout > maxCM) maxCM = (size_t)g.M() * cout;
        if (cout > maxC) maxC = cout;
    }
    for (int l = 0; l < p.L; l++) {
        p.cat[l] = take((size_t)p.geo[l].M() * 2 * p.C[l] * p.esz);
        p.pool[l] = take((size_t)p.geo[l + 1].M() * p.C[l] * p.esz);
        p.gcat[l] = take((size_t)p.geo[l].M() * 2 * p.C[l] * p.esz);
        p.gp[l] = take((size_t)p.geo[l + 1].M() * p.C[l] * p.esz);
        p.gz[l] = take((size_t)p.geo[l].M() * p.C[l] * p.esz);
    }
    p.gz[p.L] = take((size_t)p.geo[p.L].M() * p.C[p.L] * p.esz);
    p.zb = take((size_t)p.geo[p.L].M() * p.C[p.L] * p.esz);
    for (int i = 0; i < p.L; i++) {
        int l = p.L - 1 - i;
        p.zd[i] = take((size_t)p.geo[l].M() * p.C[l] * p.esz);
        p.up_mfma[i] = p.dt == MI3D_BF16 && upconv2_mfma_supported(2 * p.C[l], p.C[l], 2 * p.C[l], 2 * p.C[l]) &&
                       !mi3d_routes().force_direct;
        p.upw[i] = take(p.up_mfma[i] ? upconv2_mfma_pack_elems(2 * p.C[l], p.C[l]) * 2
                                     : upconv2_pack_floats(2 * p.C[l], p.C[l]) * sizeof(float));
        size_t wf = p.up_mfma[i] ? upconv2_mfma_bwd_ws_floats(2 * p.C[l], p.C[l], p.geo[l + 1])
                                 : upconv2_bwd_ws_floats(2 * p.C[l], p.C[l], p.geo[l + 1]);
        if (wf > wg_floats) wg_floats = wf;
    }
    p.uptmp = up_elems ? take(up_elems * p.esz) : 0;
    p.xcl = d->in_channels > 1 ? take((size_t)p.geo[0].M() * d->in_channels * p.esz) : 0;
    size_t c1 = conv1_bwd_ws_floats(p.C[0], d->out_channels);
    if (c1 > wg_floats) wg_floats = c1;
    p.sB = take(maxCM * p.esz);
    p.sB2 = take(maxCM * p.esz);
    p.sC = take(maxCM * p.esz);
    p.bnws = take(bn_ws_floats(maxC) * sizeof(float));
    p.statpart = take(statpart_floats * sizeof(float));
    p.tkcount = take((size_t)CONV3_TK_COUNTERS * sizeof(int));        // split-K ticket counters (zeroed by the forward's pack launch)
    p.skws = take(skws_floats * sizeof(float));
    p.wgws_floats = wg_floats;
    p.wgws = take(wg_floats * sizeof(float));
    p.wgws2 = take(wg_floats * sizeof(float));
    p.wgws3 = take(wg_floats * sizeof(float));       // slabs of the weight gradients on the aux stream
    // backward order of the conv layers: decoder.L-1 .. decoder.0 (level 0 first), bottleneck, encoder.L-1 .. 0; half 1 then 0
    for (int k = 0; k < 2; k++) p.flush_b[k] = p.flush_h[k] = -1;
    for (int q = 0; q < p.nblk; q++) {
        int b = 2 * p.L - q;
        for (int h = 1; h >= 0; h--) {
            int dfr = p.blk[b].h[h].defer;
            if (dfr) { p.flush_b[dfr - 1] = b; p.flush_h[dfr - 1] = h; }
        }
    }
    p.total = off;
    return 0;
}

struct Ctx {
    const Plan& p;
    char* ws;
    const void* const* params;
    hipStream_t s;
    // optional second stream: the DEFERRED weight gradients (HalfP::defer) run there, off the data-gradient chain, behind
    // at most three forks per call (events 0..2) and one join (event 3); works eagerly and inside a hipGraph capture
    hipStream_t s2 = nullptr;
    hipEvent_t* ev = nullptr;
    mutable int seq = 0;
    struct DJob { int b, h, wg_target; };
    mutable DJob dq[4 * MAXL + 2];             // weight gradients whose dy is ready and that have not been forked yet
    // forked (their event is recorded on the chain) but not yet ENQUEUED on the aux stream: the host enqueues them a few at a
    // time between the chain's next launches (drain_aux).  A step is launched by ONE host thread, and at the end of the
    // launch-bound deep-level chain it is barely ahead of the GPU: enqueueing the ten deep-level weight gradients and their slab
    // sums in one go left the chain's queue empty for ~125 us (profiles/r04_defer_eager_streams_before.txt)
    struct HJob { int b, h, wg_target, ev; };
    mutable HJob hq[4 * MAXL + 2];
    mutable int nhq = 0, hq_head = 0, waited_ev = -1;
    mutable int ndq = 0, nfork = 0;
    mutable bool aux_used = false;
    mutable bool packed = false;      // weight packs already done by the one-launch pack_all
    mutable bool tk_zeroed = false;   // this call's pack launch cleared the split-K ticket counters
    // training & 4: another forward runs beside this one on a second stream (DANN source || target): the wide BatchNorm consumers
    // (whole-CU 1024-thread workgroups) get in each other's way there (+45 us/step measured); thin consumers + finalize launches
    mutable bool beside = false;
    // one pending weight-gradient slab sum: it rides in the next BatchNorm-backward reduction launch (or is flushed
    // with its own launch when another one arrives first / at the end of the call)
    mutable SlabJob pend;
    mutable bool has_pend = false;
    // a second pending sum (reading the SECOND slab workspace): the decoder conv's sum stays pending across the transposed
    // conv's backward launch, and the next BatchNorm-backward reduction carries both
    mutable SlabJob pend2;
    mutable bool has_pend2 = false;
    bool defer_slabs = false;
    // the input gradient of a block's first conv (= the gradient of the pooled tensor one level up) left as split-K partials for
    // the MaxPool3d backward of the next segment to finish (no splitk_finish launch); pool_defer: the caller allows it
    mutable int pool_ks = 0;
    mutable bool pool_defer = false;
    mutable hipEvent_t mark_pending = nullptr;      // exchange mark waiting for the launch that completes its segment's gradients
    // called before ANY launch that writes the (single) slab workspace: an older pending sum must read it first.
    // Returns where the launcher may leave its own slab sum instead of launching it (NULL: launch immediately)
    SlabJob* pend_slot() const {
        if (has_pend) { slab_job_launch(pend, s); has_pend = false; }
        if (has_pend2) { slab_job_launch(pend2, s); has_pend2 = false; }
        if (!defer_slabs) return nullptr;
        pend = SlabJob();
        return &pend;
    }
    void pend_filled() const { has_pend = defer_slabs && pend.nblocks > 0; }
    int flush_pend() const {
        int rc = 0;
        if (has_pend) { rc = slab_job_launch(pend, s); has_pend = false; }
        if (has_pend2) { int r2 = slab_job_launch(pend2, s); has_pend2 = false; if (!rc) rc = r2; }
        return rc;
    }
    template <typename T = void> T* at(size_t off) const { return reinterpret_cast<T*>(ws + off); }
    const float* P(int i) const { return reinterpret_cast<const float*>(params[i]); }
};

// input tensor of block b: pointer, channel stride, dtype
void block_input(const Ctx& c, int b, const float* x, const void*& ptr, int& cs, int& dt) {
    const Plan& p = c.p;
    dt = p.dt;
    if (b == 0) {
        if (p.d.in_channels == 1) { ptr = x; cs = 1; dt = MI3D_F32; }
        else { ptr = c.at(p.xcl); cs = p.d.in_channels; }
    } else if (b <= p.L) { ptr = c.at(p.pool[b - 1]); cs = p.C[b - 1]; }
    else { int l = p.blk[b].level; ptr = c.at(p.cat[l]); cs = p.catcs(l); }
}
// output tensor (z2) of block b
void block_output(const Ctx& c, int b, void*& ptr, int& cs) {
    const Plan& p = c.p;
    if (b < p.L) { ptr = c.at(p.cat[b]); cs = p.catcs(b); }
    else if (b == p.L) { ptr = c.at(p.zb); cs = p.C[p.L]; }
    else { ptr = c.at(p.zd[b - p.L - 1]); cs = p.C[p.blk[b].level]; }
}

// pooled != NULL (encoder blocks on even volumes): the second apply pass also writes MaxPool3d(2,2) of the block output
int block_forward(const Ctx& c, int b, const float* x, void* const* buffers, const float* drop, int training,
                  void* pooled = nullptr, int pcs = 0) {
    const Plan& p = c.p;
    const BlockP& B = p.blk[b];
    Geo g = p.geo[B.level];
    const void* xin; int xcs, xdt;
    block_input(c, b, x, xin, xcs, xdt);
    void* zout; int zcs;
    block_output(c, b, zout, zcs);
    XfArgs xfa;                      // conv0's BatchNorm + ReLU + Dropout3d applied in conv1's staging pass (deep levels)
    for (int h = 0; h < 2; h++) {
        const HalfP& H = B.h[h];
        const void* in = h == 0 ? xin : c.at(B.z1);
        int ics = h == 0 ? xcs : H.Cout, idt = h == 0 ? xdt : p.dt;
        if (h == 1 && xfa.mode) { in = c.at(B.h[0].y); ics = B.h[0].Cout; }     // the raw conv0 output; z1 is written as a by-product
        float* rm = buffers ? (float*)buffers[H.bidx] : nullptr;
        float* rv = buffers ? (float*)buffers[H.bidx + 1] : nullptr;
        int64_t* nbt = buffers ? (int64_t*)buffers[H.bidx + 2] : nullptr;
        // training == 2: deferred running-statistics update -- buffers[bidx] is a double[2C] side buffer (ops.h bn_deferred_apply)
        const float mom = training == 2 ? -1.f : p.d.bn_momentum;
        if (training == 2) { rv = nullptr; nbt = nullptr; }
        bool fused_stats = false;
        void* zo = h == 0 ? c.at(B.z1) : zout;
        int zocs = h == 0 ? H.Cout : zcs;
        int ksd = 0, c1_blocks = 0;
        if (H.mfma) {
            if (!c.packed) MI3D_TRY(conv3_mfma_pack(c.P(H.pidx), H.Cin, H.Cout, c.at(H.wpf), c.at(H.wpd), g, c.s));
            // training: a split-K launch leaves its finishing pass to the statistics kernel (ksd = split factor).  (Deep levels
            // WITHOUT split-K -- conv with fused partial sums -> apply, two launches instead of three -- measured +0.10 ms in
            // round 2: the 8-16-chunk K loops on 32-216 workgroups cost more than the launch they save; that route is gone.)
            // round 4: a split-K launch of a training forward finishes itself behind a per-tile ticket (y, BatchNorm partial rows)
            const bool tk = training && c.tk_zeroed && !(h == 1 && xfa.mode) && conv3_mfma_ticket_ok(H.Cin, H.Cout, g);
            MI3D_TRY(conv3_mfma_fwd(in, ics, H.Cin, c.at(H.wpf), c.P(H.pidx + 1), c.at(H.y), H.Cout, H.Cout, g,
                                    training ? c.at<float>(p.statpart) : nullptr, c.at<float>(p.skws), c.s,
                                    (h == 0 && b > p.L) ? p.halves(B.level) : Halves(), Halves(), training ? &ksd : nullptr, 0, 0,
                                    (h == 1 && xfa.mode) ? &xfa : nullptr, tk ? c.at<float>(p.statpart) : nullptr,
                                    tk ? c.at<int>(p.tkcount) : nullptr));
            fused_stats = training && (tk || conv3_mfma_fuses_stats(H.Cin, H.Cout, g));
        } else if (p.dt == MI3D_BF16 && idt == MI3D_F32 && H.Cin == 1 && H.Cout % 16 == 0 && !mi3d_routes().force_direct &&
                   !mi3d_routes().no_c1_mfma) {
            // first layer on the matrix cores (taps are the K dimension), BN partial sums fused like the other convs
            MI3D_TRY(conv3_c1_fwd_mfma((const float*)in, c.P(H.pidx), c.P(H.pidx + 1), c.at(H.y), H.Cout, H.Cout, g,
                                       training ? c.at<float>(p.statpart) : nullptr, c.s));
            if (training) { fused_stats = true; c1_blocks = conv3_c1_fwd_stat_blocks(g); }
        } else {
            MI3D_TRY(conv3_direct_pack(c.P(H.pidx), H.Cin, H.Cout, c.at<float>(H.wpf), c.at<float>(H.wpd), c.s));
            MI3D_TRY(conv3_direct_fwd(idt, p.dt, in, ics, H.Cin, c.at<float>(H.wpf), c.P(H.pidx + 1), c.at(H.y), H.Cout,
                                      H.Cout, g, c.s));
        }
        int small_rows = 0;      // deep levels: the statistics' few partial rows are finished by the apply kernel (no finalize launch)
        const float* rows_at = c.at<float>(p.bnws);
        if (fused_stats) {
            // round 4: the apply pass finishes the conv epilogue's partial rows itself (bn.hip, wide consumer); the pooled pass only
            // in its two-threads-per-window form
            const int rows = c1_blocks ? c1_blocks : conv3_mfma_stat_blocks(H.Cin, H.Cout, g);
            const bool pool_pass = h == 1 && pooled;
            if (p.dt == MI3D_BF16 && bn_rows_route_ok(H.Cout, g.M(), rows) && !(pool_pass && mi3d_routes().no_pool_pair) &&
                !(c.beside && !bn_small_ok(H.Cout, g.M(), rows))) {
                small_rows = rows;
                rows_at = c.at<float>(p.statpart);
            } else
                MI3D_TRY(bn_train_finalize(c.at<float>(p.statpart), rows, H.Cout, g.M(), c.P(H.pidx + 2),
                                           c.P(H.pidx + 3), rm, rv, nbt, mom, p.d.bn_eps, c.at<float>(H.stat), c.s));
        } else if (training && ksd > 0) {
            MI3D_TRY(bn_train_stats_splitk(c.at<float>(p.skws), ksd, c.P(H.pidx + 1), c.at(H.y), H.Cout, H.Cout, g.M(), c.P(H.pidx + 2),
                                           c.P(H.pidx + 3), rm, rv, nbt, mom, p.d.bn_eps, c.at<float>(H.stat),
                                           c.at<float>(p.bnws), c.s, &small_rows));
        } else if (training) {
            MI3D_TRY(bn_train_stats(p.dt, c.at(H.y), H.Cout, H.Cout, g.M(), c.P(H.pidx + 2), c.P(H.pidx + 3), rm, rv, nbt,
                                    mom, p.d.bn_eps, c.at<float>(H.stat), c.at<float>(p.bnws), c.s, &small_rows));
        } else {
            MI3D_CHECK_ARG(rm && rv, "eval-mode forward needs running statistics");
            MI3D_TRY(bn_eval_stats(H.Cout, c.P(H.pidx + 2), c.P(H.pidx + 3), rm, rv, p.d.bn_eps, c.at<float>(H.stat), c.s));
        }
        BnSmall sm{rows_at, small_rows, c.P(H.pidx + 2), c.P(H.pidx + 3), rm, rv, nbt, mom, p.d.bn_eps};
        if (h == 0 && training && small_rows > 0 && !fused_stats && p.dt == MI3D_BF16 && B.h[1].mfma && mi3d_routes().apply_on_load &&
            conv3_mfma_xform_ok(B.h[1].Cin, B.h[1].Cout, g)) {
            // no apply launch: conv1 finishes the statistics rows in its prologue, applies while staging and writes z1 on the way
            xfa.mode = 1;
            xfa.stat = c.at<float>(H.stat); xfa.rows = c.at<float>(p.bnws); xfa.nrows = small_rows; xfa.M = g.M(); xfa.C = H.Cout;
            xfa.gamma = c.P(H.pidx + 2); xfa.beta = c.P(H.pidx + 3); xfa.rmean = rm; xfa.rvar = rv; xfa.nbt = nbt;
            xfa.momentum = mom; xfa.eps = p.d.bn_eps;
            xfa.drop = (drop && training) ? drop + H.drop_off : nullptr;
            xfa.side = (bf16*)zo; xfa.side_cs = zocs;
            continue;
        }
        if (h == 1 && pooled)
            MI3D_TRY(bn_apply_relu_drop_pool(p.dt, c.at(H.y), H.Cout, H.Cout, g, c.at<float>(H.stat),
                                             (drop && training) ? drop + H.drop_off : nullptr, zo, zocs, pooled, pcs, c.s,
                                             small_rows > 0 ? &sm : nullptr));
        else
            MI3D_TRY(bn_apply_relu_drop(p.dt, c.at(H.y), H.Cout, H.Cout, g.M(), g.V(), c.at<float>(H.stat),
                                        (drop && training) ? drop + H.drop_off : nullptr, zo, zocs, c.s, small_rows > 0 ? &sm : nullptr));
    }
    return 0;
}

// Launch the queued weight gradients on the aux stream, ordered after everything the compute stream has enqueued so far
// (their dy buffers are complete).  They run one after the other there, each followed by its slab sum, sharing the third
// slab workspace; nothing on the compute stream waits for them before the end of the step (unet_backward_impl joins).
// enqueue up to `n` forked weight gradients on the aux stream (n < 0: all).  Each runs after the fork event of its group, one
// after the other there, followed by its slab sum, sharing the third slab workspace; nothing on the compute stream waits for
// them before the end of the step (unet_backward_impl joins).
int drain_aux(const Ctx& c, const float* x, void* const* grads, int accumulate, int n) {
    const Plan& p = c.p;
    for (; c.hq_head < c.nhq && n != 0; c.hq_head++, n--) {
        const Ctx::HJob& j = c.hq[c.hq_head];
        if (j.ev != c.waited_ev) { MI3D_HIP(hipStreamWaitEvent(c.s2, c.ev[j.ev % 3], 0)); c.waited_ev = j.ev; }
        const BlockP& B = p.blk[j.b];
        const HalfP& H = B.h[j.h];
        Geo g = p.geo[B.level];
        const void* xin; int xcs, xdt;
        block_input(c, j.b, x, xin, xcs, xdt);
        const void* in = j.h == 0 ? xin : c.at(B.z1);
        int ics = j.h == 0 ? xcs : H.Cout;
        MI3D_TRY(conv3_mfma_wgrad(in, ics, H.Cin, c.at(H.dyk), H.Cout, H.Cout, g, (float*)grads[H.pidx], (float*)grads[H.pidx + 1], accumulate,
                                  c.at<float>(p.wgws3), p.wgws_floats, c.s2, (j.h == 0 && j.b > p.L) ? p.halves(B.level) : Halves(), nullptr,
                                  j.wg_target));
        c.aux_used = true;
    }
    if (c.hq_head == c.nhq) c.hq_head = c.nhq = 0;
    return 0;
}

// Fork: the queued weight gradients may start once everything the compute stream has enqueued so far is done (their dy
// buffers are complete).  lazy: only the event is recorded here, the launches are enqueued by later drain_aux calls.
int flush_deferred(const Ctx& c, const float* x, void* const* grads, int accumulate, bool lazy = false, bool force = false) {
    if (c.ndq == 0 && force) {
        MI3D_CHECK_ARG(c.nfork < 3, "flush_deferred: more than three forks in one call");
        // nothing queued (defer_mask), but the caller relies on the aux stream being ordered after this point of the chain
        // (mi3d_unet_chain_tail_blocks: its optimizer tail runs there)
        const int e = c.nfork++;
        MI3D_HIP(hipEventRecord(c.ev[e % 3], c.s));
        MI3D_HIP(hipStreamWaitEvent(c.s2, c.ev[e % 3], 0));
        c.waited_ev = e;
        c.aux_used = true;
        return 0;
    }
    if (c.ndq == 0) return 0;
    MI3D_CHECK_ARG(c.nfork < 3, "flush_deferred: more than three forks in one call");
    const int e = c.nfork++;
    MI3D_HIP(hipEventRecord(c.ev[e % 3], c.s));
    for (int q = 0; q < c.ndq; q++) c.hq[c.nhq++] = Ctx::HJob{c.dq[q].b, c.dq[q].h, c.dq[q].wg_target, e};
    c.ndq = 0;
    if (!lazy) MI3D_TRY(drain_aux(c, x, grads, accumulate, -1));
    return 0;
}

// backward of block b given dz2 (dtype T, stride dzcs); writes dxin (may be NULL) with stride dxcs
int block_backward(const Ctx& c, int b, const float* x, void* const* grads, const float* drop, const void* dz2,
                   int dzcs, void* dxin, int dxcs, int accumulate) {
    const Plan& p = c.p;
    const BlockP& B = p.blk[b];
    Geo g = p.geo[B.level];
    const void* xin; int xcs, xdt;
    block_input(c, b, x, xin, xcs, xdt);
    auto G = [&](int i) { return grads ? (float*)grads[i] : nullptr; };
    const bool aux = c.s2 != nullptr && c.ev != nullptr && !mi3d_routes().no_defer_wgrad;
    const float* dz_skp = nullptr;      // dz of half 0 left as split-K partials by half 1's fused backward launch
    int dz_ks = 0;
    for (int h = 1; h >= 0; h--) {
        const HalfP& H = B.h[h];
        int k = c.seq++;
        if (aux && c.nhq) MI3D_TRY(drain_aux(c, x, grads, accumulate, 3));      // feed the aux stream between the chain's launches
        // deferred weight gradient: dy goes to the layer's own buffer, which nobody overwrites before the aux stream has read it
        const int dbit = H.defer == 2 ? 4 : (B.level == 0 ? 1 : 2);
        const bool dfr = aux && H.defer && (mi3d_routes().defer_mask & dbit) && (G(H.pidx) || G(H.pidx + 1));
        void* dyb = dfr ? c.at(H.dyk) : c.at((k & 1) ? p.sB2 : p.sB);
        const void* dz = h == 1 ? dz2 : c.at(p.sC);
        int dcs = h == 1 ? dzcs : H.Cout;
        const void* in = h == 0 ? xin : c.at(B.z1);
        int ics = h == 0 ? xcs : H.Cout, idt = h == 0 ? xdt : p.dt;
        void* dx_f = h == 1 ? c.at(p.sC) : dxin;
        int dxs_f = h == 1 ? H.Cin : dxcs;
        // apply on load (deep levels, deferred layers): only the reduction is launched; the input-gradient conv computes dy while
        // staging (dz, y), writes it to the layer's dy buffer for the weight gradient, and publishes dgamma / dbeta
        int xf_rows = 0;
        const bool want_xf = dfr && dx_f && p.dt == MI3D_BF16 && mi3d_routes().apply_on_load && dcs % 8 == 0 &&
                             conv3_mfma_xform_ok(H.Cout, H.Cin, g) && bn_small_route(H.Cout, g.M());
        MI3D_TRY(bn_bwd(p.dt, dz, dcs, c.at(H.y), H.Cout, H.Cout, g.M(), g.V(), c.at<float>(H.stat),
                        drop ? drop + H.drop_off : nullptr, dyb, H.Cout, G(H.pidx + 2), G(H.pidx + 3), accumulate,
                        c.at<float>(p.bnws), c.s, c.has_pend ? &c.pend : (c.has_pend2 ? &c.pend2 : nullptr), h == 0 ? dz_skp : nullptr,
                        h == 0 ? dz_ks : 0, (c.has_pend && c.has_pend2) ? &c.pend2 : nullptr, want_xf ? &xf_rows : nullptr));
        c.has_pend = false;
        c.has_pend2 = false;
        if (c.mark_pending) { MI3D_HIP(hipEventRecord(c.mark_pending, c.s)); c.mark_pending = nullptr; }      // the slab sums of the marked segment rode in this launch
        if (dfr) {
            // the chain runs the input-gradient conv alone; the weight gradient is queued for the aux stream.  Its slab partition
            // is the fused launch's (conv3_mfma_bwd_wg_target), the input gradient uses the fused launch's split-K factor and the
            // same K order: both routes produce the same bits
            c.dq[c.ndq++] = Ctx::DJob{b, h, conv3_mfma_bwd_wg_target(H.Cin, H.Cout, ics, H.Cout, dx_f ? dxs_f : 8, g)};
        }
        // fork points: what is queued goes to the aux stream when the chain has finished the last layer of a group -- its
        // BatchNorm backward, or (apply on load) the input-gradient conv that writes dy -- whether or not that very layer is
        // deferred under the current defer_mask
        auto fork_here = [&]() -> int {
            if (!aux) return 0;
            // group 1 forks when the GPU is still busy with the full-resolution decoder (the host is far ahead: enqueue at once);
            // group 2 forks at the end of the launch-bound deep chain: its launches are fed in between the chain's next ones.
            // (Measured and dropped, profiles/r04_experiments_aux_wgrad.txt / _opt_tail_pool_forks.txt: one fork per layer +26 ... +43 us,
            // per deep layer only +6 us, the decoder fork one block later -1 us, another workgroup count for the aux kernels +10 ... +40 us)
            for (int q = 0; q < 2; q++)
                if (b == p.flush_b[q] && h == p.flush_h[q]) MI3D_TRY(flush_deferred(c, x, grads, accumulate, q == 1, q == 1));
            return 0;
        };
        if (xf_rows == 0) MI3D_TRY(fork_here());
        if (dfr) {
            if (dx_f) {
                const bool to_pool = h == 0 && c.pool_defer && dx_f == dxin;
                const bool defer = (h == 1 || to_pool) && c.defer_slabs && dxs_f % 8 == 0 && !mi3d_routes().no_defer_tail;
                int ksd = 0;
                XfArgs xb;
                if (xf_rows > 0) {
                    xb.mode = 2;
                    xb.y2 = (const bf16*)c.at(H.y); xb.y2cs = H.Cout;
                    xb.stat = c.at<float>(H.stat); xb.rows = c.at<float>(p.bnws); xb.nrows = xf_rows; xb.M = g.M(); xb.C = H.Cout;
                    xb.dgamma = G(H.pidx + 2); xb.dbeta = G(H.pidx + 3); xb.accumulate = accumulate;
                    xb.drop = drop ? drop + H.drop_off : nullptr;
                    xb.side = (bf16*)dyb; xb.side_cs = H.Cout;
                }
                MI3D_TRY(conv3_mfma_fwd(xf_rows > 0 ? dz : dyb, xf_rows > 0 ? dcs : H.Cout, H.Cout, c.at(H.wpd), nullptr, dx_f, dxs_f, H.Cin, g, nullptr,
                                        (dxs_f % 8 == 0) ? c.at<float>(p.skws) : nullptr, c.s, Halves(),
                                        (h == 0 && b > p.L) ? p.halves(B.level) : Halves(), defer ? &ksd : nullptr, 0, conv3_bwd_ks_target(),
                                        xf_rows > 0 ? &xb : nullptr));
                if (ksd > 0 && h == 1) { dz_skp = c.at<float>(p.skws); dz_ks = ksd; }
                if (ksd > 0 && h == 0) c.pool_ks = ksd;
            }
            if (xf_rows > 0) MI3D_TRY(fork_here());        // dy exists only now
            continue;
        }
        if (H.mfma && dx_f && (G(H.pidx) || G(H.pidx + 1)) && conv3_mfma_bwd_fused_persist_ok(H.Cin, H.Cout, ics, H.Cout, g)) {
            Halves hv = (h == 0 && b > p.L) ? p.halves(B.level) : Halves();
            SlabJob* ps = c.pend_slot();
            MI3D_TRY(conv3_mfma_bwd_fused_persist(in, ics, H.Cin, dyb, H.Cout, H.Cout, c.at(H.wpd), dx_f, dxs_f, g, G(H.pidx),
                                                  G(H.pidx + 1), accumulate, c.at<float>(p.wgws), p.wgws_floats, c.s, hv, hv, ps));
            c.pend_filled();
            continue;
        }
        if (H.mfma && dx_f && (G(H.pidx) || G(H.pidx + 1)) && !(h == 0 && b > p.L && p.planar[B.level]) &&
            conv3_mfma_bwd_fused_ok(H.Cin, H.Cout, ics, H.Cout, dxs_f, g)) {
            SlabJob* ps = c.pend_slot();
            // half 1's input gradient feeds straight into half 0's BatchNorm-backward reduction: leave a split-K result as
            // partials and let that reduction finish it (one launch less on the chain)
            const bool to_pool = h == 0 && c.pool_defer && dx_f == dxin;
            bool defer = (h == 1 || to_pool) && ps && dxs_f % 8 == 0 && !mi3d_routes().no_defer_tail;
            int ksd = 0;
            MI3D_TRY(conv3_mfma_bwd_fused(in, ics, H.Cin, dyb, H.Cout, H.Cout, c.at(H.wpd), dx_f, dxs_f, g, G(H.pidx), G(H.pidx + 1),
                                          accumulate, c.at<float>(p.wgws), p.wgws_floats, c.at<float>(p.skws), c.s, ps,
                                          defer ? &ksd : nullptr));
            if (ksd > 0 && h == 0) { c.pool_ks = ksd; ksd = 0; }
            c.pend_filled();
            if (ksd > 0) { dz_skp = c.at<float>(p.skws); dz_ks = ksd; }
            continue;
        }
        if (G(H.pidx) || G(H.pidx + 1)) {
            hipStream_t ws_ = c.s;
            float* wgws = c.at<float>(p.wgws);
            SlabJob* ps = c.pend_slot();
            if (H.mfma)
                MI3D_TRY(conv3_mfma_wgrad(in, ics, H.Cin, dyb, H.Cout, H.Cout, g, G(H.pidx), G(H.pidx + 1), accumulate,
                                          wgws, p.wgws_floats, ws_, (h == 0 && b > p.L) ? p.halves(B.level) : Halves(), ps));
            else if (p.dt == MI3D_BF16 && idt == MI3D_F32 && H.Cin == 1 && H.Cout % 16 == 0 && !mi3d_routes().force_direct)
                MI3D_TRY(conv3_mfma_wgrad_c1((const float*)in, dyb, H.Cout, H.Cout, g, G(H.pidx), G(H.pidx + 1), accumulate,
                                             wgws, p.wgws_floats, ws_, ps));
            else
                MI3D_TRY(conv3_direct_wgrad(idt, p.dt, in, ics, H.Cin, dyb, H.Cout, H.Cout, g, G(H.pidx), G(H.pidx + 1),
                                            accumulate, wgws, p.wgws_floats, ws_));
            c.pend_filled();
        }
        void* dx = h == 1 ? c.at(p.sC) : dxin;
        int dxs = h == 1 ? H.Cin : dxcs;
        if (dx) {
            if (H.mfma)
                MI3D_TRY(conv3_mfma_fwd(dyb, H.Cout, H.Cout, c.at(H.wpd), nullptr, dx, dxs, H.Cin, g, nullptr,
                                        (dxs % 8 == 0) ? c.at<float>(p.skws) : nullptr, c.s, Halves(),
                                        (h == 0 && b > p.L) ? p.halves(B.level) : Halves(), nullptr, 0, conv3_bwd_ks_target()));
            else
                MI3D_TRY(conv3_direct_fwd(p.dt, p.dt, dyb, H.Cout, H.Cout, c.at<float>(H.wpd), nullptr, dx, dxs, H.Cin, g, c.s));
        }
    }
    return 0;
}

// inference forward of block b: BatchNorm (running statistics) folded into the conv (scale in the packed filter, bias
// replaced), ReLU in the conv epilogue, output written straight to where the activated tensor lives -- no raw conv
// output, no statistics, no separate normalisation pass.  stat[0..C) = scale, stat[C..2C) = folded bias (bn_fold_all).
int block_infer(const Ctx& c, int b, const float* x) {
    const Plan& p = c.p;
    const BlockP& B = p.blk[b];
    Geo g = p.geo[B.level];
    const void* xin; int xcs, xdt;
    block_input(c, b, x, xin, xcs, xdt);
    void* zout; int zcs;
    block_output(c, b, zout, zcs);
    for (int h = 0; h < 2; h++) {
        const HalfP& H = B.h[h];
        const void* in = h == 0 ? xin : c.at(B.z1);
        int ics = h == 0 ? xcs : H.Cout, idt = h == 0 ? xdt : p.dt;
        void* zo = h == 0 ? c.at(B.z1) : zout;
        int zocs = h == 0 ? H.Cout : zcs;
        const float* scale = c.at<float>(H.stat);
        const float* fbias = scale + H.Cout;
        if (H.mfma) {
            MI3D_TRY(conv3_mfma_fwd(in, ics, H.Cin, c.at(H.wpf), fbias, zo, zocs, H.Cout, g, nullptr,
                                    (zocs % 8 == 0 && ((uintptr_t)zo % 16) == 0) ? c.at<float>(p.skws) : nullptr, c.s,
                                    (h == 0 && b > p.L) ? p.halves(B.level) : Halves(), Halves(), nullptr, 1));
        } else if (p.dt == MI3D_BF16 && idt == MI3D_F32 && H.Cin == 1 && H.Cout % 16 == 0 && !mi3d_routes().force_direct &&
                   !mi3d_routes().no_c1_mfma) {
            MI3D_TRY(conv3_c1_fwd_mfma((const float*)in, c.P(H.pidx), fbias, zo, zocs, H.Cout, g, nullptr, c.s, scale, 1));
        } else {
            MI3D_TRY(conv3_direct_pack(c.P(H.pidx), H.Cin, H.Cout, c.at<float>(H.wpf), nullptr, c.s, scale));
            MI3D_TRY(conv3_direct_fwd(idt, p.dt, in, ics, H.Cin, c.at<float>(H.wpf), fbias, zo, zocs, H.Cout, g, c.s, 1));
        }
    }
    return 0;
}

}  // namespace

extern "C" {

int mi3d_abi_version(void) { return 5; }

int mi3d_unet_num_params(const mi3d_unet_desc* d) { return d ? 8 * (2 * d->n_levels + 1) + 2 * d->n_levels + 2 : -1; }
int mi3d_unet_num_buffers(const mi3d_unet_desc* d) { return d ? 6 * (2 * d->n_levels + 1) : -1; }
int mi3d_unet_num_segments(const mi3d_unet_desc* d) { return d ? 2 * d->n_levels + 2 : -1; }

size_t mi3d_unet_workspace_bytes(const mi3d_unet_desc* d) {
    Plan p;
    if (build_plan(d, p) != 0) return 0;
    return p.total;
}

int64_t mi3d_unet_dropout_count(const mi3d_unet_desc* d) {
    Plan p;
    if (build_plan(d, p) != 0) return -1;
    const HalfP& last = p.blk[p.nblk - 1].h[1];
    return last.drop_off + (int64_t)d->N * last.Cout;
}

int mi3d_unet_segment_params(const mi3d_unet_desc* d, int seg, int* r) {
    Plan p;
    MI3D_TRY(build_plan(d, p));
    int L = p.L;
    MI3D_CHECK_ARG(seg >= 0 && seg < 2 * L + 2 && r, "bad segment %d", seg);
    r[2] = r[3] = -1;
    if (seg == 0) { r[0] = p.final_pidx(); r[1] = r[0] + 2; }
    else if (seg <= L) {
        int i = L - seg;          // decoder.i and upconvs.i
        r[0] = p.blk[L + 1 + i].h[0].pidx; r[1] = r[0] + 8;
        r[2] = p.up_pidx(i); r[3] = r[2] + 2;
    } else if (seg == L + 1) { r[0] = p.blk[L].h[0].pidx; r[1] = r[0] + 8; }
    else { int l = 2 * L + 1 - seg; r[0] = p.blk[l].h[0].pidx; r[1] = r[0] + 8; }
    return 0;
}

}  // extern "C"

// the segmentation loss of the training loop folded into the 1x1x1 head (head_loss.hip): forward half / backward half
struct HeadLoss {
    const int64_t* labels; LossCfg cfg; float* loss_out; float* coef; float* metrics_out; void* loss_ws; void* met_ws;   // forward
    const float* grad_scale;                                                                                              // backward
    const float* teacher;                                                                 // (N,C,V) teacher logits iff cfg.w_kd != 0
};
static LossCfg cfg_of(const mi3d_loss_cfg* c) {
    LossCfg k;
    k.w_ce = c->w_ce; k.region_kind = c->region_kind; k.w_reg = c->w_reg; k.alpha = c->alpha; k.beta = c->beta; k.eps = c->eps;
    k.w_kd = c->w_kd; k.temp = c->temperature > 0.f ? c->temperature : 1.f;
    return k;
}

// MFMA weight images of the DoubleConv blocks [b0, b1) (+ the transposed convs) of the training forward / backward, one launch
static int pack_training_weights(const Ctx& c, int b0, int b1, bool upconvs, bool zero_tickets = false) {
    const Plan& p = c.p;
    PackJobs J;
    J.n = 0; J.nblocks = 0;
    if (zero_tickets) { J.zero = c.at<int>(p.tkcount); J.nzero = CONV3_TK_COUNTERS; }
    for (int b = b0; b < b1 && b < p.nblk; b++)
        for (int h = 0; h < 2; h++) {
            const HalfP& H = p.blk[b].h[h];
            if (H.mfma) MI3D_TRY(pack_all_add_conv3(J, c.P(H.pidx), H.Cin, H.Cout, c.at(H.wpf), c.at(H.wpd), p.geo[p.blk[b].level]));
        }
    if (upconvs)
        for (int i = 0; i < p.L; i++) {
            int l = p.L - 1 - i;
            if (p.up_mfma[i]) MI3D_TRY(pack_all_add_upconv(J, c.P(p.up_pidx(i)), 2 * p.C[l], p.C[l], c.at(p.upw[i])));
        }
    if (zero_tickets && J.n > 0) c.tk_zeroed = true;      // (pack_all_launch launches nothing for an empty job list)
    return pack_all_launch(J, c.s);
}

extern "C" int mi3d_unet_pack_from(const mi3d_unet_desc* d, const void* const* params, void* workspace, size_t workspace_bytes,
                                   int first_block, void* stream) {
    Plan p;
    MI3D_TRY(build_plan(d, p));
    MI3D_CHECK_ARG(params && workspace && workspace_bytes >= p.total && first_block >= 0, "mi3d_unet_pack_from: bad arguments");
    Ctx c{p, (char*)workspace, params, (hipStream_t)stream};
    return pack_training_weights(c, first_block, p.nblk, true);
}

// leading encoder blocks whose weight gradients stay on the data-gradient chain (no deferred layer), i.e. are produced last
extern "C" int mi3d_unet_chain_tail_blocks(const mi3d_unet_desc* d) {
    Plan p;
    if (!d || build_plan(d, p) != 0) return 0;
    if (mi3d_routes().no_defer_wgrad || p.flush_b[1] < 0) return 0;      // no fork behind the deep levels: the aux stream is not ordered
    int k = 0;
    while (k < p.L && p.blk[k].h[0].defer == 0 && p.blk[k].h[1].defer == 0) k++;
    return (k < p.L) ? k : 0;
}

static int unet_forward_impl(const mi3d_unet_desc* d, const float* x, const void* const* params, void* const* buffers,
                      const float* drop_scales, int training, float* logits, float* gap_out, void* workspace,
                      size_t workspace_bytes, void* stream, const HeadLoss* hl) {
    Plan p;
    MI3D_TRY(build_plan(d, p));
    MI3D_CHECK_ARG(x && params && workspace, "mi3d_unet_forward: null pointer");
    MI3D_CHECK_ARG(workspace_bytes >= p.total, "workspace too small: %zu < %zu", workspace_bytes, p.total);
    MI3D_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    Ctx c{p, (char*)workspace, params, (hipStream_t)stream};
    c.beside = (training & 4) != 0;
    training &= 3;
    int L = p.L;
    if (d->in_channels > 1)
        MI3D_TRY(ncdhw_to_ndhwc(p.dt, x, c.at(p.xcl), d->in_channels, d->in_channels, d->N, p.geo[0].V(), c.s));
    // every MFMA weight pack of the network in one launch (prepacked_from = k > 0: the caller's mi3d_unet_pack_from already did
    // the blocks >= k and the transposed convs behind its optimizer update)
    MI3D_TRY(pack_training_weights(c, 0, d->prepacked_from > 0 ? d->prepacked_from : p.nblk, d->prepacked_from <= 0, training != 0));
    c.packed = true;
    for (int l = 0; l < L; l++) {
        // fused apply + pool: even sides (every voxel in exactly one window) and 32-bit element indices
        const bool even = p.geo[l].D % 2 == 0 && p.geo[l].H % 2 == 0 && p.geo[l].W % 2 == 0 &&
                          p.geo[l].M() * p.C[l] < (1ll << 31) && !mi3d_routes().no_pool_fuse;
        MI3D_TRY(block_forward(c, l, x, buffers, drop_scales, training, even ? c.at(p.pool[l]) : nullptr, p.C[l]));
        if (!even) MI3D_TRY(maxpool2_fwd(p.dt, c.at(p.cat[l]), p.catcs(l), p.C[l], p.geo[l], c.at(p.pool[l]), p.C[l], c.s));
    }
    MI3D_TRY(block_forward(c, L, x, buffers, drop_scales, training));
    if (gap_out) MI3D_TRY(gap_fwd(p.dt, c.at(p.zb), p.C[L], p.C[L], d->N, p.geo[L].V(), gap_out, c.s));
    for (int i = 0; i < L; i++) {
        int l = L - 1 - i;
        float* wf = c.at<float>(p.upw[i]);
        float* wb = wf + (size_t)cdiv(p.C[l], 8) * (2 * p.C[l]) * 64;
        const void* uin = i == 0 ? c.at(p.zb) : c.at(p.zd[i - 1]);
        char* catl = c.at<char>(p.cat[l]);
        void* udst = p.resize[l] ? c.at(p.uptmp) : (void*)(catl + p.half_off(l));
        int udcs = p.resize[l] ? p.C[l] : p.catcs(l);
        if (p.up_mfma[i]) {
            if (!c.packed) MI3D_TRY(upconv2_mfma_pack(c.P(p.up_pidx(i)), 2 * p.C[l], p.C[l], c.at(p.upw[i]), c.s));
            MI3D_TRY(upconv2_mfma_fwd(uin, 2 * p.C[l], 2 * p.C[l], c.at(p.upw[i]), c.P(p.up_pidx(i) + 1),
                                      udst, udcs, p.C[l], p.geo[l + 1], c.s));
        } else {
            MI3D_TRY(upconv2_pack(c.P(p.up_pidx(i)), 2 * p.C[l], p.C[l], wf, wb, c.s));
            MI3D_TRY(upconv2_fwd(p.dt, uin, 2 * p.C[l], 2 * p.C[l], wf, c.P(p.up_pidx(i) + 1), udst, udcs, p.C[l], p.geo[l + 1], c.s));
        }
        if (p.resize[l])
            MI3D_TRY(nearest_resize_fwd(p.dt, udst, udcs, p.C[l], p.up_geo(l), catl + p.half_off(l), p.catcs(l), p.geo[l], c.s));
        MI3D_TRY(block_forward(c, L + 1 + i, x, buffers, drop_scales, training));
    }
    if (hl) {
        MI3D_CHECK_ARG(head_loss_ok(p.dt, c.at(p.zd[L - 1]), p.C[0], p.C[0], d->out_channels, hl->cfg),
                       "mi3d_unet_forward_loss: no fused head + loss for this configuration (see mi3d_unet_head_loss_supported)");
        return head_loss_fwd(c.at(p.zd[L - 1]), p.C[0], p.C[0], c.P(p.final_pidx()), c.P(p.final_pidx() + 1), hl->labels, hl->teacher,
                             d->N, d->out_channels, p.geo[0].V(), hl->cfg, hl->loss_out, hl->coef, hl->loss_ws, c.s, d->D,
                             hl->metrics_out, hl->met_ws, logits);
    }
    if (logits)       // NULL: the caller does not want the logits (DANN target pass) or runs the head with mi3d_unet_head_loss_forward
        MI3D_TRY(conv1_fwd(p.dt, c.at(p.zd[L - 1]), p.C[0], p.C[0], c.P(p.final_pidx()), c.P(p.final_pidx() + 1), logits,
                           d->out_channels, d->N, p.geo[0].V(), c.s));
    return 0;
}

extern "C" {

// head + loss on the decoder output the last mi3d_unet_forward / mi3d_unet_infer (logits = NULL) left in this workspace
int mi3d_unet_head_loss_forward(const mi3d_unet_desc* d, const void* const* params, const int64_t* labels, const float* teacher_logits,
                                const mi3d_loss_cfg* cfg, float* loss_out, float* coef, float* metrics_out, void* loss_workspace,
                                void* metrics_workspace, float* logits_opt, void* workspace, size_t workspace_bytes, void* stream) {
    Plan p;
    MI3D_TRY(build_plan(d, p));
    MI3D_CHECK_ARG(params && labels && cfg && loss_out && coef && loss_workspace && workspace, "mi3d_unet_head_loss_forward: null pointer");
    MI3D_CHECK_ARG(workspace_bytes >= p.total, "workspace too small: %zu < %zu", workspace_bytes, p.total);
    Ctx c{p, (char*)workspace, params, (hipStream_t)stream};
    LossCfg k = cfg_of(cfg);
    MI3D_CHECK_ARG(head_loss_ok(p.dt, c.at(p.zd[p.L - 1]), p.C[0], p.C[0], d->out_channels, k),
                   "mi3d_unet_head_loss_forward: no fused head + loss for this configuration (see mi3d_unet_head_loss_supported)");
    return head_loss_fwd(c.at(p.zd[p.L - 1]), p.C[0], p.C[0], c.P(p.final_pidx()), c.P(p.final_pidx() + 1), labels, teacher_logits, d->N,
                         d->out_channels, p.geo[0].V(), k, loss_out, coef, loss_workspace, c.s, d->D, metrics_out, metrics_workspace,
                         logits_opt);
}

int mi3d_unet_forward(const mi3d_unet_desc* d, const float* x, const void* const* params, void* const* buffers,
                      const float* drop_scales, int training, float* logits, float* gap_out, void* workspace,
                      size_t workspace_bytes, void* stream) {
    return unet_forward_impl(d, x, params, buffers, drop_scales, training, logits, gap_out, workspace, workspace_bytes, stream, nullptr);
}

int mi3d_unet_head_loss_supported(const mi3d_unet_desc* d, const mi3d_loss_cfg* cfg) {
    Plan p;
    if (!d || !cfg || build_plan(d, p) != 0) return 0;
    LossCfg k = cfg_of(cfg);
    // the plan's activations are 256-byte aligned and dense: only dtype / channel counts / loss terms decide
    return head_loss_bwd_ok(p.dt, nullptr, p.C[0], p.C[0], d->out_channels, k, nullptr, p.C[0]) ? 1 : 0;
}

int mi3d_unet_forward_loss(const mi3d_unet_desc* d, const float* x, const void* const* params, void* const* buffers,
                           const float* drop_scales, int training, const int64_t* labels, const float* teacher_logits,
                           const mi3d_loss_cfg* cfg, float* loss_out, float* coef, float* metrics_out, void* loss_workspace,
                           void* metrics_workspace, float* logits_opt, float* gap_out, void* workspace, size_t workspace_bytes,
                           void* stream) {
    MI3D_CHECK_ARG(labels && cfg && loss_out && coef && loss_workspace, "mi3d_unet_forward_loss: null pointer");
    HeadLoss hl{labels, cfg_of(cfg), loss_out, coef, metrics_out, loss_workspace, metrics_workspace, nullptr, teacher_logits};
    return unet_forward_impl(d, x, params, buffers, drop_scales, training, logits_opt, gap_out, workspace, workspace_bytes, stream, &hl);
}

int mi3d_unet_bn_apply_deferred(const mi3d_unet_desc* d, void* const* buffers, const void* const* side, void* stream) {
    Plan p;
    MI3D_TRY(build_plan(d, p));
    MI3D_CHECK_ARG(buffers && side, "mi3d_unet_bn_apply_deferred: null pointer");
    BnDeferJobs J;
    J.n = 0; J.momentum = d->bn_momentum;
    for (int b = 0; b < p.nblk; b++)
        for (int h = 0; h < 2; h++) {
            const HalfP& H = p.blk[b].h[h];
            MI3D_CHECK_ARG(buffers[H.bidx] && buffers[H.bidx + 1] && side[H.bidx], "mi3d_unet_bn_apply_deferred: missing buffer %d", H.bidx);
            J.j[J.n++] = BnDeferJob{(float*)buffers[H.bidx], (float*)buffers[H.bidx + 1], (int64_t*)buffers[H.bidx + 2],
                                    (const double*)side[H.bidx], H.Cout};
        }
    return bn_deferred_apply(J, (hipStream_t)stream);
}

int mi3d_unet_infer(const mi3d_unet_desc* d, const float* x, const void* const* params, void* const* buffers,
                    float* logits, float* gap_out, void* workspace, size_t workspace_bytes, void* stream) {
    Plan p;
    MI3D_TRY(build_plan(d, p));
    MI3D_CHECK_ARG(x && params && buffers && workspace, "mi3d_unet_infer: null pointer");
    MI3D_CHECK_ARG(workspace_bytes >= p.total, "workspace too small: %zu < %zu", workspace_bytes, p.total);
    MI3D_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    Ctx c{p, (char*)workspace, params, (hipStream_t)stream};
    int L = p.L;
    if (d->in_channels > 1)
        MI3D_TRY(ncdhw_to_ndhwc(p.dt, x, c.at(p.xcl), d->in_channels, d->in_channels, d->N, p.geo[0].V(), c.s));
    {   // launch 1: every BatchNorm folded; launch 2: every MFMA weight pack, BatchNorm scale multiplied in
        BnFoldJobs F;
        F.n = 0; F.eps = d->bn_eps;
        PackJobs J;
        J.n = 0; J.nblocks = 0;
        for (int b = 0; b < p.nblk; b++)
            for (int h = 0; h < 2; h++) {
                const HalfP& H = p.blk[b].h[h];
                MI3D_CHECK_ARG(buffers[H.bidx] && buffers[H.bidx + 1], "mi3d_unet_infer needs running statistics");
                float* scale = c.at<float>(H.stat);
                F.j[F.n++] = BnFoldJob{c.P(H.pidx + 2), c.P(H.pidx + 3), (const float*)buffers[H.bidx], (const float*)buffers[H.bidx + 1],
                                       c.P(H.pidx + 1), scale, scale + H.Cout, H.Cout};
                if (H.mfma) MI3D_TRY(pack_all_add_conv3(J, c.P(H.pidx), H.Cin, H.Cout, c.at(H.wpf), c.at(H.wpd), p.geo[p.blk[b].level], scale));
            }
        for (int i = 0; i < L; i++) {
            int l = L - 1 - i;
            if (p.up_mfma[i]) MI3D_TRY(pack_all_add_upconv(J, c.P(p.up_pidx(i)), 2 * p.C[l], p.C[l], c.at(p.upw[i])));
        }
        MI3D_TRY(bn_fold_all(F, c.s));
        MI3D_TRY(pack_all_launch(J, c.s));
        c.packed = true;
    }
    for (int l = 0; l < L; l++) {
        MI3D_TRY(block_infer(c, l, x));
        MI3D_TRY(maxpool2_fwd(p.dt, c.at(p.cat[l]), p.catcs(l), p.C[l], p.geo[l], c.at(p.pool[l]), p.C[l], c.s));
    }
    MI3D_TRY(block_infer(c, L, x));
    if (gap_out) MI3D_TRY(gap_fwd(p.dt, c.at(p.zb), p.C[L], p.C[L], d->N, p.geo[L].V(), gap_out, c.s));
    for (int i = 0; i < L; i++) {
        int l = L - 1 - i;
        float* wf = c.at<float>(p.upw[i]);
        float* wb = wf + (size_t)cdiv(p.C[l], 8) * (2 * p.C[l]) * 64;
        const void* uin = i == 0 ? c.at(p.zb) : c.at(p.zd[i - 1]);
        char* catl = c.at<char>(p.cat[l]);
        void* udst = p.resize[l] ? c.at(p.uptmp) : (void*)(catl + p.half_off(l));
        int udcs = p.resize[l] ? p.C[l] : p.catcs(l);
        if (p.up_mfma[i]) {
            MI3D_TRY(upconv2_mfma_fwd(uin, 2 * p.C[l], 2 * p.C[l], c.at(p.upw[i]), c.P(p.up_pidx(i) + 1),
                                      udst, udcs, p.C[l], p.geo[l + 1], c.s));
        } else {
            MI3D_TRY(upconv2_pack(c.P(p.up_pidx(i)), 2 * p.C[l], p.C[l], wf, wb, c.s));
            MI3D_TRY(upconv2_fwd(p.dt, uin, 2 * p.C[l], 2 * p.C[l], wf, c.P(p.up_pidx(i) + 1), udst, udcs, p.C[l], p.geo[l + 1], c.s));
        }
        if (p.resize[l])
            MI3D_TRY(nearest_resize_fwd(p.dt, udst, udcs, p.C[l], p.up_geo(l), catl + p.half_off(l), p.catcs(l), p.geo[l], c.s));
        MI3D_TRY(block_infer(c, L + 1 + i, x));
    }
    if (logits)
        MI3D_TRY(conv1_fwd(p.dt, c.at(p.zd[L - 1]), p.C[0], p.C[0], c.P(p.final_pidx()), c.P(p.final_pidx() + 1), logits,
                           d->out_channels, d->N, p.geo[0].V(), c.s));
    return 0;
}

}  // extern "C"

// Exchange marks (data parallel): "record this event once every gradient of the segments <= seg is complete".  Set by
// mi3d_unet_backward_marks for the NEXT backward call of the calling thread; that call stays ONE run of launches (a weight-gradient
// slab sum of segment seg rides in the first BatchNorm-backward reduction of segment seg + 1, and the event is recorded right
// behind that launch) instead of being cut into two calls at the exchange.
struct BwdMarks { int n = 0; int seg[4]; hipEvent_t ev[4]; };
static thread_local BwdMarks g_marks;
extern "C" int mi3d_unet_backward_marks(const int* segs, void* const* events, int n) {
    MI3D_CHECK_ARG(n >= 0 && n <= 4 && (n == 0 || (segs && events)), "mi3d_unet_backward_marks: at most 4 marks");
    g_marks.n = n;
    for (int i = 0; i < n; i++) { g_marks.seg[i] = segs[i]; g_marks.ev[i] = (hipEvent_t)events[i]; }
    return 0;
}
extern "C" int mi3d_stream_wait_event(void* stream, void* event) {
    MI3D_CHECK_ARG(event, "mi3d_stream_wait_event: null event");
    MI3D_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return 0;
}

static int unet_backward_impl(const mi3d_unet_desc* d, const float* x, const void* const* params, void* const* grads,
                       const float* drop_scales, const float* dlogits_in, const float* dgap, float gap_scale, int accumulate,
                       int seg_begin, int seg_end, void* workspace, size_t workspace_bytes, void* stream, void* aux_stream,
                       void* const* events, int aux_join, const HeadLoss* hl) {
    Plan p;
    MI3D_TRY(build_plan(d, p));
    MI3D_CHECK_ARG(x && params && grads && workspace, "mi3d_unet_backward: null pointer");
    MI3D_CHECK_ARG(workspace_bytes >= p.total, "workspace too small: %zu < %zu", workspace_bytes, p.total);
    // `dlogits` below only says "the segmentation branch has a gradient": with the fused head it is never dereferenced
    const float* dlogits = hl ? reinterpret_cast<const float*>(hl->coef) : dlogits_in;
    MI3D_CHECK_ARG(dlogits || dgap, "mi3d_unet_backward: neither dlogits nor dgap given");
    int L = p.L, nseg = 2 * L + 2;
    MI3D_CHECK_ARG(seg_begin >= 0 && seg_end <= nseg && seg_begin <= seg_end, "bad segment range [%d,%d)", seg_begin, seg_end);
    Ctx c{p, (char*)workspace, params, (hipStream_t)stream};
    if (aux_stream && events) { c.s2 = (hipStream_t)aux_stream; c.ev = (hipEvent_t*)events; }
    float* wgws = c.at<float>(p.wgws);
    c.defer_slabs = !mi3d_routes().no_pend_slabs;
    auto G = [&](int i) { return (float*)grads[i]; };
    BwdMarks marks = g_marks;
    g_marks.n = 0;
    for (int seg = seg_begin; seg < seg_end; seg++) {
        if (seg > seg_begin)
            for (int i = 0; i < marks.n; i++)
                if (marks.seg[i] == seg - 1) {
                    // two marks on one launch cannot happen (one event per segment); an older mark still pending means the
                    // segment in between launched no BatchNorm backward: complete it now
                    if (c.mark_pending) { MI3D_TRY(c.flush_pend()); MI3D_HIP(hipEventRecord(c.mark_pending, c.s)); }
                    c.mark_pending = marks.ev[i];
                }
        if (seg == 0) {
            if (!dlogits) continue;
            SlabJob* ps = c.pend_slot();
            if (hl) {
                MI3D_CHECK_ARG(head_loss_bwd_ok(p.dt, c.at(p.zd[L - 1]), p.C[0], p.C[0], d->out_channels, hl->cfg, c.at(p.gz[0]), p.C[0]),
                               "mi3d_unet_backward_loss: no fused head + loss for this configuration (see mi3d_unet_head_loss_supported)");
                MI3D_TRY(head_loss_bwd(c.at(p.zd[L - 1]), p.C[0], p.C[0], c.P(p.final_pidx()), c.P(p.final_pidx() + 1), hl->labels,
                                       hl->teacher, d->out_channels, hl->cfg, hl->coef, hl->grad_scale, c.at(p.gz[0]), p.C[0], G(p.final_pidx()),
                                       G(p.final_pidx() + 1), accumulate, wgws, d->N, p.geo[0].V(), c.s, ps));
            } else
                MI3D_TRY(conv1_bwd(p.dt, c.at(p.zd[L - 1]), p.C[0], p.C[0], c.P(p.final_pidx()), dlogits, d->out_channels,
                                   c.at(p.gz[0]), p.C[0], G(p.final_pidx()), G(p.final_pidx() + 1), accumulate, wgws, d->N,
                                   p.geo[0].V(), c.s, ps));
            c.pend_filled();
        } else if (seg <= L) {
            if (!dlogits) continue;
            int l = seg - 1, i = L - 1 - l;       // decoder.i works at level l
            MI3D_TRY(block_backward(c, L + 1 + i, x, grads, drop_scales, c.at(p.gz[l]), p.C[l], c.at(p.gcat[l]), p.catcs(l), accumulate));
            const void* uin = i == 0 ? c.at(p.zb) : c.at(p.zd[i - 1]);
            float* wf = c.at<float>(p.upw[i]);
            float* wb = wf + (size_t)cdiv(p.C[l], 8) * (2 * p.C[l]) * 64;
            char* gcatl = c.at<char>(p.gcat[l]);
            const void* gup = gcatl + p.half_off(l);
            int gupcs = p.catcs(l);
            if (p.resize[l]) {       // adjoint of the nearest resize in front of the concat (models/unet.py:81-83)
                MI3D_TRY(nearest_resize_bwd(p.dt, gup, gupcs, p.C[l], p.geo[l], c.at(p.uptmp), p.C[l], p.up_geo(l), c.s));
                gup = c.at(p.uptmp); gupcs = p.C[l];
            }
            // the decoder conv's pending slab sum (it reads wgws) stays pending across the transposed conv's backward, which
            // therefore writes its slabs to the second workspace; the next BatchNorm-backward reduction carries both sums: one
            // chain link less per level (not with the two-stream weight gradients, which own that workspace)
            const bool keep = p.up_mfma[i] && c.has_pend && !c.has_pend2 && c.defer_slabs && !mi3d_routes().no_upbwd_carry;
            if (p.up_mfma[i] && keep) {
                c.pend2 = SlabJob();
                MI3D_TRY(upconv2_mfma_bwd(uin, 2 * p.C[l], 2 * p.C[l], gup, gupcs, p.C[l],
                                          c.at(p.upw[i]), c.at(p.gz[l + 1]), 2 * p.C[l], G(p.up_pidx(i)), G(p.up_pidx(i) + 1),
                                          accumulate, c.at<float>(p.wgws2), p.wgws_floats, p.geo[l + 1], c.s, &c.pend2));
                c.has_pend2 = c.pend2.nblocks > 0;
                continue;
            }
            SlabJob* ps = c.pend_slot();
            if (p.up_mfma[i]) {
                MI3D_TRY(upconv2_mfma_bwd(uin, 2 * p.C[l], 2 * p.C[l], gup, gupcs, p.C[l],
                                          c.at(p.upw[i]), c.at(p.gz[l + 1]), 2 * p.C[l], G(p.up_pidx(i)), G(p.up_pidx(i) + 1),
                                          accumulate, wgws, p.wgws_floats, p.geo[l + 1], c.s, ps));
                c.pend_filled();
            } else
                MI3D_TRY(upconv2_bwd(p.dt, uin, 2 * p.C[l], 2 * p.C[l], gup, gupcs, p.C[l], wb,
                                     c.at(p.gz[l + 1]), 2 * p.C[l], G(p.up_pidx(i)), G(p.up_pidx(i) + 1), accumulate, wgws,
                                     p.wgws_floats, p.geo[l + 1], c.s));
        } else if (seg == L + 1) {
            if (dgap)
                MI3D_TRY(gap_bwd(p.dt, dgap, gap_scale, c.at(p.gz[L]), p.C[L], p.C[L], d->N, p.geo[L].V(), dlogits ? 1 : 0, c.s));
            // (the gradient of the pooled tensor may stay split-K partials when the MaxPool3d backward that reads it is launched
            // by this very call: a later call would not know about them)
            c.pool_defer = seg + 1 < seg_end && !mi3d_routes().no_pool_splitk;
            MI3D_TRY(block_backward(c, L, x, grads, drop_scales, c.at(p.gz[L]), p.C[L], c.at(p.gp[L - 1]), p.C[L - 1], accumulate));
            c.pool_defer = false;
        } else {
            int l = 2 * L + 1 - seg;              // encoder.l
            MI3D_TRY(maxpool2_bwd(p.dt, c.at(p.gp[l]), p.C[l], c.at(p.cat[l]), p.catcs(l), dlogits ? c.at(p.gcat[l]) : nullptr,
                                  p.catcs(l), c.at(p.gz[l]), p.C[l], p.C[l], p.geo[l], c.s,
                                  c.pool_ks > 0 ? c.at<float>(p.skws) : nullptr, c.pool_ks));
            c.pool_ks = 0;
            void* dx = l > 0 ? c.at(p.gp[l - 1]) : nullptr;
            c.pool_defer = l > 0 && seg + 1 < seg_end && !mi3d_routes().no_pool_splitk;
            MI3D_TRY(block_backward(c, l, x, grads, drop_scales, c.at(p.gz[l]), p.C[l], dx, l > 0 ? p.C[l - 1] : 0, accumulate));
            c.pool_defer = false;
        }
    }
    MI3D_TRY(c.flush_pend());
    if (c.mark_pending) { MI3D_HIP(hipEventRecord(c.mark_pending, c.s)); c.mark_pending = nullptr; }
    for (int i = 0; i < marks.n; i++)
        if (marks.seg[i] == seg_end - 1) MI3D_HIP(hipEventRecord(marks.ev[i], c.s));
    // weight gradients still queued (a call that ends before the group's own fork point): they go out now, so that every
    // gradient of the segments [seg_begin, seg_end) is at least in flight when the call returns
    if (c.s2 && c.ev) {
        MI3D_TRY(flush_deferred(c, x, grads, accumulate));
        MI3D_TRY(drain_aux(c, x, grads, accumulate, -1));
    }
    if (c.aux_used) {
        // event 3 = "the aux stream has finished what this call gave it".  aux_join: the compute stream waits for it here, i.e.
        // everything is ordered before whatever the caller enqueues next on `stream`; otherwise the CALLER orders its consumers
        // (optimizer, gradient exchange, the end of a graph capture) after event 3 / the aux stream
        MI3D_HIP(hipEventRecord(c.ev[3], c.s2));
        if (aux_join) MI3D_HIP(hipStreamWaitEvent(c.s, c.ev[3], 0));
    }
    return 0;
}

extern "C" {

int mi3d_unet_backward(const mi3d_unet_desc* d, const float* x, const void* const* params, void* const* grads,
                       const float* drop_scales, const float* dlogits, const float* dgap, float gap_scale, int accumulate,
                       int seg_begin, int seg_end, void* workspace, size_t workspace_bytes, void* stream, void* aux_stream,
                       void* const* events, int aux_join) {
    return unet_backward_impl(d, x, params, grads, drop_scales, dlogits, dgap, gap_scale, accumulate, seg_begin, seg_end, workspace,
                              workspace_bytes, stream, aux_stream, events, aux_join, nullptr);
}

int mi3d_unet_backward_loss(const mi3d_unet_desc* d, const float* x, const void* const* params, void* const* grads,
                            const float* drop_scales, const int64_t* labels, const float* teacher_logits, const mi3d_loss_cfg* cfg,
                            const float* coef, const float* grad_scale, const float* dgap, float gap_scale, int accumulate, int seg_begin,
                            int seg_end, void* workspace, size_t workspace_bytes, void* stream, void* aux_stream, void* const* events,
                            int aux_join) {
    MI3D_CHECK_ARG(labels && cfg && coef, "mi3d_unet_backward_loss: null pointer");
    HeadLoss hl{labels, cfg_of(cfg), nullptr, const_cast<float*>(coef), nullptr, nullptr, nullptr, grad_scale, teacher_logits};
    return unet_backward_impl(d, x, params, grads, drop_scales, nullptr, dgap, gap_scale, accumulate, seg_begin, seg_end, workspace,
                              workspace_bytes, stream, aux_stream, events, aux_join, &hl);
}

}  // extern "C"
