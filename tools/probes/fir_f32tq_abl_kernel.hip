// tools/probes/fir_f32tq_abl_kernel.hip -- the ABLATION copy of minimal-sdr_amd/csrc/msdr_fir_f32tq.hiph (round 5: the product kernel no
// longer carries these switches; this file is included by tools/probes/fir_tq_bench.hip only and must follow the product kernel by hand).
// ---- the product header's text from here on, with `ABL` ----
// msdr_fir_f32tq.hiph -- arm_fir_f32 (prototype src/CMSIS_5/arm_math.h:1182-1186; CMSIS-DSP 1.5.x semantics
// y[n] = sum_k pCoeffs[k] x[n-(N-1)+k]): the taps-in-registers kernel of msdr_fir_f32tr.hiph with the TILES DEALT
// FROM A QUEUE IN ADDRESS ORDER instead of one long stream per wave (gfx950, round 3).
//
// Why.  The stage is bound by its stream (4 B in + 4 B out per sample), and what a 1:1 read / write stream reaches on this part depends
// on the ORDER in which the chip touches memory (tools/probes/copy_ceiling_probe.hip, copy_order_probe.hip, fir_stream_probe.hip;
// profiles/r03/stream_order.md): a plain one-shot copy, whose workgroups are dispatched in address order, sweeps memory as one compact
// front and moves 0.77-0.82 of 8 TB/s; the same accesses in a scrambled order 0.63; 2048 long-lived waves that each walk their own far-
// apart (channel, segment) stream -- the shape of round 2's kernel on this layout -- 0.65; long-lived waves that DRAW their next tile from a counter
// in address order, at this kernel's residency, 0.76-0.77.  A FIR tile needs nothing from the wave's previous tile but the halo, and
// that is re-read (1 KB per 4 KB tile, found in the L2 of the same XCD: the neighbouring tile was fetched a moment ago by a wave
// of the same front), so the tiles can be dealt out in any order.
//
// How.  The flattened tile space (channel-major, 1024 outputs per tile) is cut into F contiguous FRONTS; workgroup b serves front
// b % F (F a multiple of 8: the workgroups with equal b % 8 share an XCD under round-robin placement -- for speed only, nothing depends
// on it) and each of its waves draws one tile at a time from the front's counter (one returning atomic per tile, issued one tile
// ahead of its use; counters 64 bytes apart, F = 32 keeps each below the ~88 draws per microsecond one word takes).  Every tile
// fetches its whole window -- H halo samples and the 1024 of the tile: 5 x 16 bytes per lane at 256 taps -- so the block-floating-
// point scale of a window is taken from exactly the samples in it, and the halo copy inside LDS is gone.  The software pipeline inside
// the wave is the one of msdr_fir_f32tr.hiph: while the products of tile A issue, the same instruction stream carries the stores of
// the parked tile Z, the conversion of the fetched tile B into the other window buffer and the fetch of the drawn tile C.  Tiles that
// open a row (halo from the carried history), end it (partial) or lie in an unaligned row take the same steps one after the other with
// bounds checks ("cold").  The two counter sets alternate between launches: a launch zeroes the set the next one will use.
// Arithmetic, tap tables, window layout and parking are msdr_fir_f32tr.hiph's.  Algorithmic bytes: 4 in + 4 out per sample.
#pragma once
#include "msdr_fir_f32tq.hiph"      // TqParams, the load / store policy, the layout
#define MSDR_TQ_ABL_COPY 1

namespace msdr {

// ABL (diagnostic instantiations of tools/probes/fir_tq_bench.hip only; the library builds ABL = 0): 1 = no matrix instructions,
// 2 = no global loads / stores in the steady state, 4 = tiles assigned round-robin inside the front instead of drawn,
// 8 = 30 of the 36 product triples of a 9-step tile (what a 2 x 2 block-Toeplitz "fast FIR" split would issue; results meaningless),
// 16 = the side work such a split adds: half as many conversions again (the pre-added third stream) and a third fragment pair per two k-steps
template <int NS, bool SKIP1, int ABL = 0>
__global__ __launch_bounds__(256, 2) void fir_f32tq_abl_kernel(const TqParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int H = 32 * (NS - 1);
    constexpr int AS = tr_array_bytes(NS);
    constexpr int WB = 2 * AS;                                               // one window buffer: hi | lo
    constexpr int NSL = NS + 1;                                              // k-steps of the products = slices of the side work
    constexpr int NCH = (H + kTrTile) / 4;                                   // 16-byte chunks of samples in a window
    constexpr int NLD = (NCH + 63) / 64;                                     // ... per lane
    constexpr bool kWholeLoads = (NLD * 64 == NCH);                          // else the last load of a lane is partial (lanes < NCH - 64 (NLD - 1))
    // Round 4.  With H = 256 the halo is exactly the window's first load (64 chunks) and a tile's LAST load is exactly the halo of the tile
    // behind it, lane for lane: a wave that goes on to the next tile of the same row keeps those four registers instead of fetching the
    // 1 KB again (traffic / algorithmic 1.049 -> 1.01 with runs of 4; a fifth of the load instructions gone).
    constexpr bool kReuse = kWholeLoads && H == 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *win0 = smem + (size_t)wave * tr_wave_bytes(NS);           // window 0 | window 1 | parking area
    unsigned char *park = win0 + 2 * WB;
    const MSDR_CAS F32TrHeader *hdr = (const MSDR_CAS F32TrHeader *)p.tab;
    const int ex = hdr->ex, fixed_k = hdr->fixed_k;
    const bool use_fixed = hdr->use_fixed != 0;
    const long long n = p.n;
    if (blockIdx.x == 0 && tid < kTqMaxFronts) p.ctr_next[tid * kTqCtrStride] = 0u;

    const unsigned front = blockIdx.x % p.fronts;
    const unsigned fbase = front * p.per_front;
    const unsigned fcount = fbase >= p.total ? 0u : (p.total - fbase < p.per_front ? p.total - fbase : p.per_front);
    unsigned *ctr = p.ctr + front * kTqCtrStride;
    if (fcount == 0u) return;

    // ---- every tap of the filter, as A fragments: th/tl[family][step] ----
    f16x8 th[2][NS], tl[2][NS];
    {
        const char *tp = p.tab + kTrHdrBytes + lane * 16;
#pragma unroll
        for (int F = 0; F < 2; F++)
#pragma unroll
            for (int s = 0; s < NS; s++) {
                th[F][s] = *reinterpret_cast<const f16x8 *>(tp + ((F * NS + s) * 2) * 1024);
                tl[F][s] = *reinterpret_cast<const f16x8 *>(tp + ((F * NS + s) * 2 + 1) * 1024);
            }
    }
    // lane constants
    const int xoff = (lane & 15) * kTrBlk + (lane >> 4) * 16;                // X fragment of k-step 0 (column lane & 15, k group lane >> 4)
    const int pwoff = (lane & 15) * kTrParkPitch + (lane >> 4) * 16;         // parking: column lane & 15, rows 4 (lane >> 4) ..
    const int proff = (lane >> 4) * kTrParkPitch + (lane & 15) * 16;         // reading back in time order: chunk lane (+ 64 k: 4 k columns on)
    const int csoff = tr_win_off(4 * lane);                                  // staging: window samples 4 lane .. (+ 256 j: 4 blocks on)

    // ---- the tile queue ----
    // A draw is one returning atomic add by lane 0.  Written as inline assembly: left to the compiler, an atomicAdd under `if (lane == 0)`
    // goes through its wave-aggregation pass (mbcnt / bcnt / readfirstlane) and is waited for on the spot (s_waitcnt vmcnt(0) right behind
    // it: the whole memory pipeline of the wave drained once per tile).  Here the draw is ISSUED at the top of a tile and TAKEN at its
    // end, inside the same iteration (no register holding an unfinished draw lives across a loop edge, where a compiler-made copy
    // would read it too early).  The counter of vector-memory operations retires in issue order: s_waitcnt vmcnt(N) returns once at
    // most N of them are outstanding, so it proves the draw has returned only if AT LEAST N vector-memory instructions were issued behind
    // the draw on every path to the take -- `younger` must never exceed that number (a smaller value is always safe: it waits for some of
    // the younger operations too).  Only instructions the wave issues unconditionally count: a load under a lane condition sits behind an
    // s_cbranch_execz in the binary.  tests/test_tq_draw_registers.py walks every path of the product binary and checks the arithmetic.
    // A draw hands out a run of `run` consecutive tiles; inside a run the wave goes on to the next tile without asking.  The atomic is
    // issued (and taken) on every tile all the same -- with an increment of 0 inside a run -- so that the code between a draw and its take
    // has ONE shape for the disassembly test to walk.
    const unsigned run = 1u << p.run_shift, nruns = (fcount + run - 1) >> p.run_shift;
    auto run_more = [&](unsigned idx) { return idx != kTqNone && (((idx - fbase) & (run - 1u)) != run - 1u) && idx + 1u < fbase + fcount; };
    auto draw_issue = [&](unsigned &gp, unsigned inc) {
        unsigned long long save;
        asm volatile("s_and_saveexec_b64 %1, 1\n\t"
                     "global_atomic_add %0, %2, %3, %4 sc0\n\t"
                     "s_mov_b64 exec, %1"
                     : "=&v"(gp), "=&s"(save) : "v"(0u), "v"(inc), "s"(ctr) : "memory", "scc");       // (s_and_saveexec writes SCC)
    };
    auto draw_take = [&](unsigned gp, auto younger_tag) -> unsigned {
        unsigned g;
        asm volatile("s_waitcnt vmcnt(%2)\n\t"
                     "v_readfirstlane_b32 %0, %1\n\t"
                     "s_nop 4"                                            // (a VALU-written SGPR in front of whatever reads it next)
                     : "=s"(g) : "v"(gp), "n"(decltype(younger_tag)::value) : "memory");
        return g < nruns ? fbase + (g << p.run_shift) : kTqNone;
    };
    // (ABL & 4: round-robin instead -- wave r of the front's waves takes tiles r, r + R, r + 2 R ...)
    const unsigned rr_step = ((gridDim.x - front + p.fronts - 1) / p.fronts) * 4u;
    unsigned rr_next = (blockIdx.x / p.fronts) * 4u + (unsigned)wave;
    auto draw_sync = [&]() -> unsigned {
        if constexpr (ABL & 4) { const unsigned g = rr_next; rr_next += rr_step; return g < fcount ? fbase + g : kTqNone; }
        unsigned gp; draw_issue(gp, 1u); return draw_take(gp, std::integral_constant<int, 0>{});
    };
    auto next_sync = [&](unsigned idx) -> unsigned { return run_more(idx) ? idx + 1u : draw_sync(); };      // the tile behind idx in this wave's sequence
    // tile idx -> (row offset in samples, first sample inside the row, "full": complete and aligned -- its outputs may leave as four
    // 1 KB stores --, "hot": full and its halo inside the row -- its window may be streamed)
    auto geo = [&](unsigned idx, long long &row, long long &t0, bool &full, bool &hot) {
        const unsigned ch = p.tpr_shift >= 0 ? idx >> p.tpr_shift : idx / p.tpr;
        const unsigned k = idx - ch * p.tpr;
        row = (long long)ch * n; t0 = (long long)k * kTrTile;
        const bool al = p.all_aligned || (((reinterpret_cast<uintptr_t>(p.x + row) | reinterpret_cast<uintptr_t>(p.y + row)) & 15) == 0);
        full = al && t0 + kTrTile <= n;
        hot = full && k > 0;
    };

    // ---- the pieces of one tile ----
    f32x4 pre[NLD];                          // raw samples of the fetched tile: chunk lane + 64 j = window samples 4 (lane + 64 j) ..+3
#pragma unroll
    for (int j = 0; j < NLD; j++) pre[j] = (f32x4)(0.0f);
    const bool last_valid = kWholeLoads || lane < NCH - 64 * (NLD - 1);
    auto fetch_chunk = [&](long long row, long long t0, int j) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(p.x + row + t0 - H) + lane;
        if (j < NLD - 1 || kWholeLoads || last_valid) pre[j] = MSDR_TQ_LOAD(src + 64 * j);
    };
    auto fetch_hot = [&](long long row, long long t0) {
#pragma unroll
        for (int j = 0; j < NLD; j++) fetch_chunk(row, t0, j);
    };
    int kcur = fixed_k, knew = 0;            // the current / the next window's samples are stored as x 2^k
    auto scale_next = [&]() {
        float mf = 0.0f;
#pragma unroll
        for (int j = 0; j < NLD; j++) mf = tr_absmax4(mf, pre[j]);           // (chunks past the window stay zero)
        const unsigned m = fm_wave_max(__float_as_uint(mf));
        const int kd = fm_scale_exp(m);
        knew = use_fixed ? fixed_k : kd;
    };
    auto convert = [&](unsigned char *wn, int j) {
        if (j < NLD - 1 || kWholeLoads || last_valid) tr_write4_at(wn + csoff + 4 * j * kTrBlk, AS, pre[j], fm_pow2(knew));
    };
    // cold staging of tile (ch, t0) into window w, bounds-checked (row start: the carried history; row end: zeros); returns its scale
    auto stage_cold = [&](unsigned char *w, unsigned idx) -> int {
        const unsigned ch = p.tpr_shift >= 0 ? idx >> p.tpr_shift : idx / p.tpr;
        const long long t0 = (long long)(idx - ch * p.tpr) * kTrTile;
        const float *__restrict__ xrow = p.x + (long long)ch * n;
        const float *__restrict__ hrow = p.hist + (long long)ch * p.hist_len;
        const bool al = (reinterpret_cast<uintptr_t>(xrow) & 15) == 0;
        auto fetch4 = [&](long long t) -> f32x4 {
            if (al && t >= 0 && t + 4 <= n) return *reinterpret_cast<const f32x4 *>(xrow + t);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = fm_load(xrow, hrow, t + e, n, p.hist_len);
            return v;
        };
        int k = fixed_k;
        if (!use_fixed) {
            unsigned m = 0u;
            for (int u = lane; u < NCH; u += 64) m = fm_absmax4(m, fetch4(t0 - H + 4LL * u));
            k = fm_scale_exp(fm_wave_max(m));
        }
        const float pre_s = fm_pow2(k);
        for (int u = lane; u < NCH; u += 64) tr_write4(w, AS, 4 * u, fetch4(t0 - H + 4LL * u), pre_s);
        return k;
    };

    f32x4 acc[2][2];                         // [j][F]: sub-tile a0 = 32 j + 16 F
    f16x8 xh, xl;
    auto x_first = [&](const unsigned char *w) {
        xh = *reinterpret_cast<const f16x8 *>(w + xoff); xl = *reinterpret_cast<const f16x8 *>(w + xoff + AS);
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int F = 0; F < 2; F++) acc[j][F] = (f32x4)(0.0f);
    };
    // products of k-step sx (X fragments = window samples 64 b + 32 sx + 8 g ..+7); the fragments of step sx + 1 are requested first
    auto x_step = [&](const unsigned char *w, auto sx_tag) {
        constexpr int sx = decltype(sx_tag)::value;
        f16x8 nh = xh, nl = xl;
        if constexpr (sx < NS) {
            constexpr int off = ((sx + 1) >> 1) * kTrBlk + ((sx + 1) & 1) * 64;
            nh = *reinterpret_cast<const f16x8 *>(w + xoff + off); nl = *reinterpret_cast<const f16x8 *>(w + xoff + AS + off);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int s = sx - j;
            if (s < 0 || s >= NS) continue;
#pragma unroll
            for (int F = 0; F < 2; F++) {
                if (SKIP1 && F == 1 && s == 0) continue;
                if ((ABL & 8) && j == 1 && F == 1 && s < 6) continue;
                if constexpr (ABL & 1) { acc[j][F] += (f32x4){(float)xl[0], (float)xh[1], (float)th[F][s][0], (float)tl[F][s][1]}; continue; }
                acc[j][F] = __builtin_amdgcn_mfma_f32_16x16x32_f16(th[F][s], xl, acc[j][F], 0, 0, 0);
                acc[j][F] = __builtin_amdgcn_mfma_f32_16x16x32_f16(th[F][s], xh, acc[j][F], 0, 0, 0);
                acc[j][F] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tl[F][s], xh, acc[j][F], 0, 0, 0);
            }
        }
        if constexpr ((ABL & 16) != 0 && (sx & 1) == 0 && sx < NS) {       // a third stream's fragments: read and folded in so that they stay
            constexpr int off2 = ((sx + 2) >> 1) * kTrBlk;
            const f16x8 eh = *reinterpret_cast<const f16x8 *>(w + xoff + off2), el = *reinterpret_cast<const f16x8 *>(w + xoff + AS + off2);
            acc[0][0][0] += (float)eh[0] + (float)el[0];
        }
        xh = nh; xl = nl;
    };
    constexpr auto n_products = [](int sx) {            // matrix instructions of k-step sx
        int c = 0;
        for (int j = 0; j < 2; j++) { const int s = sx - j; if (s < 0 || s >= NS) continue; for (int F = 0; F < 2; F++) if (!(SKIP1 && F == 1 && s == 0)) c += 3; }
        return c;
    };
    // park the finished tile: lane (column b, row group g) holds outputs 64 b + a0 + 4 g ..+3 of each sub-tile
    auto park_write = [&]() {
        const float post = fm_pow2(-kcur - ex);
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int F = 0; F < 2; F++) *reinterpret_cast<f32x4 *>(park + pwoff + 4 * (32 * j + 16 * F)) = acc[j][F] * post;
    };
    auto park_read = [&](int k) -> f32x4 { return *reinterpret_cast<const f32x4 *>(park + proff + 4 * k * kTrParkPitch); };   // 16-byte chunk lane + 64 k of the tile
    auto store_full = [&](float *o, int k, f32x4 v) { MSDR_TQ_STORE(v, reinterpret_cast<f32x4 *>(o) + lane + 64 * k); };
    auto store_checked = [&](long long row, long long t) {
        float *orow = p.y + row;
        const bool al = (reinterpret_cast<uintptr_t>(orow) & 15) == 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const f32x4 v = park_read(k);
            const int c = lane + 64 * k;
            float *dst = orow + t + 4 * c;
            const long long rem = n - t - 4 * c;
            if (rem >= 4 && al) *reinterpret_cast<f32x4 *>(dst) = v;
            else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (e < rem) dst[e] = v[e];
            }
        }
    };

    // ---- the tile loop.  In flight: Z = parked outputs, A = the window being multiplied, B = fetched (hot) or to be staged (cold),
    // C = drawn, to be fetched next; D = the draw of this iteration ----
    const unsigned first = draw_sync();
    if (first == kTqNone) return;
    long long arow, at0; bool afull, ahot_unused;
    geo(first, arow, at0, afull, ahot_unused);
    int cur = 0;
    kcur = stage_cold(win0, first);
    unsigned bidx = next_sync(first);
    long long brow = 0, bt0 = 0; bool bfull = false, bhot = false;
    if (bidx != kTqNone) geo(bidx, brow, bt0, bfull, bhot);
    bool have_pre = bidx != kTqNone && bhot;             // pre = the raw samples of B
    if (have_pre) fetch_hot(brow, bt0);
    unsigned cidx = bidx != kTqNone ? next_sync(bidx) : kTqNone;
    long long crow = 0, ct0 = 0; bool cfull = false, chot = false;
    if (cidx != kTqNone) geo(cidx, crow, ct0, cfull, chot);
    bool zvalid = false, zfull = false;
    long long zrow = 0, zt0 = 0;
    bool avalid = true;
    while (avalid) {
        // (an inner loop of its own: the registers that live across iterations -- the fetched samples above all -- then keep their
        //  places on the back edge, see msdr_fir_f32tr.hiph)
        while (zvalid && zfull && afull && have_pre) {
            const unsigned char *w = win0 + cur * WB;
            unsigned char *wn = win0 + (cur ^ 1) * WB;
            // ======== steady state: Z is parked, A is in window `cur` (both complete and aligned: a partial tile stored as a whole one
            // would run into the next row), B is complete, aligned, its halo inside the row, and in pre ========
            float *zo = p.y + zrow + zt0;
            const long long frow = chot ? crow : brow, ft0 = chot ? ct0 : bt0;      // nothing to stream next: B again (valid addresses), unused
            unsigned didx = kTqNone, gp = 0u; long long drow = 0, dt0 = 0; bool dfull = false, dhot = false;
            const bool cont = run_more(cidx);                                       // D is C's neighbour: no new run
            const bool creuse = kReuse && chot && cidx == bidx + 1u;                // C is B's neighbour in the same row: its halo is B's last load, in pre[NLD - 1]
            f32x4 pv[4];
            x_first(w);
            // Side work, piece i.  The draw of the tile after C goes out first; then the stores of the parked tile Z (first k-steps), the
            // next window (scale, conversion of B's samples, fetched one iteration earlier), last the fetch of C into the same registers,
            // and the draw is taken.  (Measured and not kept: converting B at the start of the iteration and re-requesting every 16-byte
            // chunk at once for C, so that a wave has loads in flight nearly all the time -- 6 % slower, in every ablation: more
            // bytes in flight cost bandwidth on this part, profiles/r03/stream_order.md.)
            auto side = [&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value;
                if constexpr (i == 0) { if constexpr (!(ABL & 4)) draw_issue(gp, cont ? 0u : 1u); pv[0] = park_read(0); pv[1] = park_read(1); }
                if constexpr (i == 1) { if constexpr (!(ABL & 2)) { store_full(zo, 0, pv[0]); store_full(zo, 1, pv[1]); } pv[2] = park_read(2); pv[3] = park_read(3); }
                if constexpr (i == 2) { if constexpr (!(ABL & 2)) { store_full(zo, 2, pv[2]); store_full(zo, 3, pv[3]); } else { pre[0] += pv[0] + pv[1] + pv[2] + pv[3]; } }
                if constexpr (i == 3) { scale_next(); }
                if constexpr (i == 4) { convert(wn, 0); convert(wn, 1); }
                if constexpr (i == 5) {
#pragma unroll
                    for (int j = 2; j < NLD; j++) convert(wn, j);
                    if constexpr (ABL & 16) {          // the pre-added stream: (NLD + 1) / 2 more chunks converted (x0 + x1, its own split), written over the parked tile
#pragma unroll
                        for (int j = 0; j < (NLD + 1) / 2; j++) tr_write4_at(park + (lane & 15) * 16 + 256 * j, 2048, pre[2 * j] + pre[(2 * j + 1) % NLD], fm_pow2(knew));
                    }
                }
                if constexpr (i == 6) {
                    if constexpr (!(ABL & 2)) {
                        if constexpr (kReuse) {
                            if (creuse) {
                                pre[0] = pre[NLD - 1];
#pragma unroll
                                for (int j = 1; j < NLD; j++) fetch_chunk(frow, ft0, j);
                            } else fetch_hot(frow, ft0);
                        } else fetch_hot(frow, ft0);
                    }
                    if constexpr (ABL & 4) { didx = rr_next < fcount ? fbase + rr_next : kTqNone; rr_next += rr_step; }
                    else {
                        // younger than the draw on every path: Z's four stores, C's loads (a partial last load is conditional, the halo load is skipped inside a run: not counted)
                        didx = draw_take(gp, std::integral_constant<int, (ABL & 2) ? 0 : NLD + 4 - (kWholeLoads ? 0 : 1) - (kReuse ? 1 : 0)>{});
                        if (cont) didx = cidx + 1u;
                    }
                    if (didx != kTqNone) geo(didx, drow, dt0, dfull, dhot);
                }
            };
            constexpr auto slice_of = [](int i) { return i < 3 ? (i < NSL ? i : NSL - 1) : (NSL - 4 + (i - 3) > 0 ? NSL - 4 + (i - 3) : 0); };
            auto slice = [&](auto sx_tag) {
                constexpr int sx = decltype(sx_tag)::value;
                x_step(w, sx_tag);
                if constexpr (slice_of(0) == sx) side(std::integral_constant<int, 0>{});
                if constexpr (slice_of(1) == sx) side(std::integral_constant<int, 1>{});
                if constexpr (slice_of(2) == sx) side(std::integral_constant<int, 2>{});
                if constexpr (slice_of(3) == sx) side(std::integral_constant<int, 3>{});
                if constexpr (slice_of(4) == sx) side(std::integral_constant<int, 4>{});
                if constexpr (slice_of(5) == sx) side(std::integral_constant<int, 5>{});
                if constexpr (slice_of(6) == sx) side(std::integral_constant<int, 6>{});
#pragma unroll
                for (int q = 0; q < n_products(sx); q++) MSDR_TR_SGB_PAIR();
                __builtin_amdgcn_sched_barrier(0);
            };
            tr_static_for<NSL>(slice);
            park_write();
            zrow = arow; zt0 = at0;                      // (A was full: zfull stays true)
            arow = brow; at0 = bt0; afull = true;        // (B was hot)
            bidx = cidx; brow = crow; bt0 = ct0; bfull = cfull; bhot = chot;
            have_pre = chot;                             // (C cold or none: what was fetched is B again, not used)
            cidx = didx; crow = drow; ct0 = dt0; cfull = dfull; chot = dhot;
            kcur = knew; cur ^= 1;
        }
        // ======== the same steps one after the other, with every check ========
        const unsigned char *w = win0 + cur * WB;
        unsigned char *wn = win0 + (cur ^ 1) * WB;
        unsigned gp = 0u;
        const bool cont_cold = run_more(cidx);
        if constexpr (!(ABL & 4)) draw_issue(gp, cont_cold ? 0u : 1u);        // (unconditional, as its take below: once the queue is dry a draw comes back empty)
        if (zvalid) {
            if (zfull) {
                float *zo = p.y + zrow + zt0;
#pragma unroll
                for (int k = 0; k < 4; k++) store_full(zo, k, park_read(k));
            } else store_checked(zrow, zt0);
        }
        x_first(w);
        tr_static_for<NSL>([&](auto sx_tag) { x_step(w, sx_tag); });
        park_write();
        zvalid = true; zfull = afull; zrow = arow; zt0 = at0;
        unsigned didx;
        if constexpr (ABL & 4) didx = draw_sync(); else { didx = draw_take(gp, std::integral_constant<int, 0>{}); if (cont_cold) didx = cidx + 1u; }
        if (bidx != kTqNone) {
            if (have_pre) {
                scale_next();
#pragma unroll
                for (int j = 0; j < NLD; j++) convert(wn, j);
                kcur = knew;
            } else kcur = stage_cold(wn, bidx);
            cur ^= 1;
            arow = brow; at0 = bt0; afull = bfull;
            bidx = cidx; brow = crow; bt0 = ct0; bfull = cfull; bhot = chot;
            have_pre = bidx != kTqNone && bhot;
            if (have_pre) fetch_hot(brow, bt0);
            cidx = didx; cfull = false; chot = false;
            if (cidx != kTqNone) geo(cidx, crow, ct0, cfull, chot);
        } else avalid = false;
    }
    if (zvalid) {
        if (zfull) {
            float *zo = p.y + zrow + zt0;
#pragma unroll
            for (int k = 0; k < 4; k++) store_full(zo, k, park_read(k));
        } else store_checked(zrow, zt0);
    }
}

}  // namespace msdr
