// tests/cpp/geometry_check.hip -- host-only checks of the LDS layout arithmetic the kernels and the table builders share (the __host__ __device__
// helpers of the kernel headers, compiled as they are; nothing is launched: runs without a GPU).  tests/test_layout_geometry.py builds and runs it.
#include <cstdio>
#include <set>
#include "../../minimal-sdr_amd/csrc/msdr_chain_mfw.hiph"
#include "../../minimal-sdr_amd/csrc/msdr_chain_mfb.hiph"
#include "../../minimal-sdr_amd/csrc/msdr_chain_q15mb.hiph"

using namespace msdr;

static int bad = 0;
#define CHECK(cond, ...) do { if (!(cond)) { bad++; printf("FAIL %s: ", #cond); printf(__VA_ARGS__); printf("\n"); } } while (0)

int main()
{
    // ---- the compact B operand of the full-rate layout (mw_compact_stride): room for one k-step past the last chunk, 16-byte aligned, and
    // the 16 lanes one LDS pass serves (8 shifted copies x 2 offsets) in 16 different 16-byte slots of the 256-byte bank space ----
    for (int H = 32; H <= 2048; H += 32) {
        const int cs = mw_compact_stride(H);
        CHECK(cs % 16 == 0, "H %d cs %d", H, cs);
        CHECK(cs >= 2 * (H + 72) + 16, "H %d cs %d: the prefetch one step past the last chunk leaves the copy", H, cs);
        CHECK(cs < 2 * H + 160 + 256, "H %d cs %d: more padding than the bank rule needs", H, cs);
        for (int khalf = 0; khalf < 2; khalf++)
            for (int b0 = 0; b0 < 32; b0 += 16) {
                std::set<int> slots;
                for (int b = b0; b < b0 + 16; b++) {
                    const int lane = b + 32 * khalf;
                    const int off = ((7 - lane) & 7) * cs + 16 * ((lane >> 5) + ((31 - (lane & 31)) >> 3));      // the kernel's blane_off
                    CHECK(off % 16 == 0, "H %d lane %d", H, lane);
                    slots.insert((off % 256) / 16);
                }
                CHECK(slots.size() == 16, "H %d k-half %d columns %d..: %zu distinct slots", H, khalf, b0, slots.size());
            }
        // every entry a lane reads lies inside its copy: u0 = 16 j + 8 h + 31 - b for chunks j = 0 .. (H + 32) / 16 (one past the last)
        const int J = (H + 32) / 16;
        for (int lane = 0; lane < 64; lane++) {
            const int b = lane & 31, h = lane >> 5, s = (7 - b) & 7;
            const int u0 = 16 * J + 8 * h + 31 - b;
            CHECK(u0 % 8 == s, "H %d lane %d", H, lane);
            CHECK(2 * (u0 - s) + 16 <= cs, "H %d lane %d: entry %d past the copy (%d bytes)", H, lane, u0 - s + 7, cs);
        }
    }
    // ---- the full-rate layout with the compact B operand keeps long filters on the matrix cores: two filters' copies beside at least two waves'
    // windows in 160 KB for every tap count the tests claim (the host refuses the layout below two waves per workgroup) ----
    for (int taps = 128; taps <= 1021; taps++)
        for (int S = 0; S <= 2; S++) {
            const int H = mf_halo(taps + 2 * S);
            const int bsteps = (2 * 16 * mw_compact_stride(H) + 2047) / 2048;
            int w = 0;
            while (w < 16 && mw_lds_bytes(H, bsteps, w + 1, true) <= 160 * 1024) w++;
            CHECK(w >= 2, "taps %d sections %d: %d waves fit", taps, S, w);
            if (taps <= 516) CHECK(w >= 6, "taps %d sections %d: %d waves fit (six at 512 taps: profiles/r05/nco_long_taps.txt)", taps, S, w);
            // (the k-step fragments of the same filters: 2 KB per 16 window samples and filter)
            const int frag_steps = 2 * ((H + 32) / 16);
            if (taps >= 512) CHECK(mw_lds_bytes(H, frag_steps, 2, true) > 160 * 1024, "taps %d: the fragment layout would fit two waves after all", taps);
        }
    // ---- block-cadence geometry (msdr_chain_mfb.hiph): tiles of 32 rows = CPT channels x RPC rows for every block length the host accepts ----
    for (int n : {32, 64, 128, 256, 512}) {
        CHECK(mb_n_ok(n), "n %d", n);
        CHECK(mb_rpc(n) * mb_cpt(n) == 32 && mb_rpc(n) * 32 == n, "n %d rpc %d cpt %d", n, mb_rpc(n), mb_cpt(n));
        for (int H = 32; H <= 2048; H += 32) {
            CHECK(mb_rw(H, n) * 32 >= H + n, "H %d n %d rw %d", H, n, mb_rw(H, n));
            CHECK(mb_lds_bytes(H, n, 4, 1, 1) > (size_t)4 * 2048, "H %d n %d", H, n);
            CHECK(mb_lds_bytes(H, n, 4, 2, 1) - mb_lds_bytes(H, n, 4, 1, 1) == mb_wave_bytes(H, n, 1), "H %d n %d: a wave's share", H, n);
            if (H <= 512) {
                CHECK(qb_chan_bytes(H, n) * 2 == H + n, "H %d n %d", H, n);
                CHECK(qb_nodes_lds_bytes(H, n, 4, 4) == qb_lds_bytes(H, n, 4, 4, 1) + (size_t)2 * 4 * 8 * kTqbPitch * 4, "H %d n %d", H, n);
            }
        }
    }
    for (int n : {0, 16, 96, 100, 1024}) CHECK(!mb_n_ok(n), "n %d", n);
    // the swizzle of a window row's four 16-byte chunks: a permutation for every row address, chunk c and c ^ 2 (hi / lo pieces) 32 bytes apart
    for (unsigned row = 0; row < 64 * 1024; row += 64)
        for (unsigned c = 0; c < 2; c++) {
            const unsigned a = mb_chunk(row, c);
            CHECK((a & ~63u) == row && (a & 15u) == 0, "row %u c %u -> %u", row, c, a);
            CHECK(((a ^ 32u) & ~63u) == row, "row %u", row);
            CHECK(mb_chunk(row, c ^ 1) != a && (mb_chunk(row, c ^ 1) ^ 32u) != a, "row %u c %u", row, c);
        }
    printf("geometry_check: %d failures\n", bad);
    return bad ? 1 : 0;
}
