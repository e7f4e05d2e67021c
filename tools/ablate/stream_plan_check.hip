// stream_plan_check.hip -- host-side check of the stream GEMM's work partition (no GPU needed: only the host half
// of gemm_x3.h runs).  For every shape on the command line (or the built-in list) it plans the launch exactly as
// winograd.hip does, walks every workgroup's range with the SAME function the kernel uses (x3_walk_next) and checks
//   * every piece has an even number >= 4 of K stages inside its tile, its tile lies in its slot's range;
//   * the pieces of a (frequency, tile) cover stages [0, S) exactly once, in workgroup order;
//   * a workgroup parks at most one piece (its first) and the workgroup that continues it is the next non-empty one;
//   * operand block indices stay inside the operand.
// usage: stream_plan_check [M N K seg_mode seg_len seg_skip nw]...   prints "ok <n shapes>" or the first violation.
//   hipcc -O1 -std=c++17 --offload-arch=gfx950 -DX3_STREAM_TOOL -I ot-gan_amd/csrc -I tools/ablate tools/ablate/stream_plan_check.hip -o tools/ablate/bin/stream_plan_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <vector>

#include "gemm_x3.h"
#include "gemm_x3_stream.h"     // the stream-K kernel (left the library in round 5); build with -DX3_STREAM_TOOL

void otgan_set_error(const char*, ...) {}
void otgan_prof_begin(int, double, double, hipStream_t) {}
void otgan_prof_end(int, hipStream_t) {}

struct Shape {
  int M, N, K, seg_mode, seg_len, seg_skip, nw;
};

#define FAIL(...) do { printf("FAIL M=%d N=%d K=%d seg=%d/%d/%d nw=%d: ", sh.M, sh.N, sh.K, sh.seg_mode, sh.seg_len, sh.seg_skip, sh.nw); printf(__VA_ARGS__); printf("\n"); return false; } while (0)

static bool check(const Shape& sh, long* pieces_total, long* parked_total) {
  BgArgs b;
  memset(&b, 0, sizeof(b));
  b.M = sh.M; b.N = sh.N; b.K = sh.K;
  b.seg_mode = sh.seg_mode; b.seg_len = sh.seg_len; b.seg_skip = sh.seg_skip;
  b.tiles_m = (sh.M + X3_BM - 1) / X3_BM;
  b.tiles_n = (sh.N + X3_BN - 1) / X3_BN;
  x3_build_fmap(b);
  if (!x3_plan_stream(b, sh.nw)) {
    printf("refused M=%d N=%d K=%d seg=%d nw=%d (one-tile kernel)\n", sh.M, sh.N, sh.K, sh.seg_mode, sh.nw);
    return true;
  }
  const int tiles = b.tiles_m * b.tiles_n;
  std::map<long, std::vector<std::pair<int, int>>> cover;   // (f, tile) -> pieces in workgroup order (ascending wg)
  for (int xcd = 0; xcd < 8; ++xcd) {
    int prev_nonempty = -1;
    unsigned prev_stop = 0;
    for (int wg = 0; wg < sh.nw; ++wg) {
      const unsigned pstart = b.sk_bound[xcd][wg], pstop = b.sk_bound[xcd][wg + 1];
      if (pstop < pstart) FAIL("xcd %d wg %d: bounds decrease", xcd, wg);
      if (pstart >= pstop) continue;
      if (prev_nonempty >= 0 && prev_stop != pstart) FAIL("xcd %d wg %d: gap after workgroup %d", xcd, wg, prev_nonempty);
      X3Walk w;
      x3_walk_begin(b, xcd, pstop, w);
      int n = 0, f, tile, s0, s1, S;
      unsigned lo;
      std::vector<std::pair<long, std::pair<int, int>>> mine;
      while (x3_walk_next(b, xcd, pstart, w, f, tile, s0, s1, S, lo)) {
        if (++n > X3_SK_MAXPIECES) FAIL("xcd %d wg %d: more than %d pieces", xcd, wg, X3_SK_MAXPIECES);
        if (w.slot < 0 || w.slot >= 8 || b.fmap[xcd][w.slot] == -1) FAIL("xcd %d wg %d: slot %d", xcd, wg, w.slot);
        if (f < 0 || f >= kWF) FAIL("xcd %d wg %d: frequency %d", xcd, wg, f);
        if (tile < w.t0 || tile >= w.t1 || tile >= tiles) FAIL("xcd %d wg %d: tile %d outside [%d, %d)", xcd, wg, tile, w.t0, w.t1);
        if (s0 < 0 || s1 > S || s1 - s0 < 4 || ((s1 - s0) & 1) || (s0 & 1)) FAIL("xcd %d wg %d: piece [%d, %d) of %d", xcd, wg, s0, s1, S);
        if (S != x3_item_stages(b, f, tile)) FAIL("stage count");
        if (n > 1 && s1 != S) FAIL("xcd %d wg %d: piece %d does not end its tile", xcd, wg, n);
        if (n == 1 && s1 != S) ++*parked_total;
        // k blocks of the piece
        for (int j = s0; j < s1; ++j) {
          int kb = j;
          if (sh.seg_mode == 1) {
            int lo0 = 0, len0 = 0, lo1 = 0, cc = 0, nrun = 0;
            while (cc < 4) {
              if (!s2_present(cc, f, sh.seg_skip)) { ++cc; continue; }
              int e = cc + 1;
              while (e < 4 && s2_present(e, f, sh.seg_skip)) ++e;
              if (nrun == 0) { lo0 = cc * sh.seg_len; len0 = (e - cc) * sh.seg_len; } else { lo1 = cc * sh.seg_len; }
              ++nrun;
              cc = e;
            }
            const int steps0 = len0 / X3_SK;
            kb = j < steps0 ? lo0 / X3_SK + j : lo1 / X3_SK - steps0 + j;
          }
          if (kb < 0 || kb >= sh.K / X3_SK) FAIL("xcd %d wg %d: k block %d of %d", xcd, wg, kb, sh.K / X3_SK);
        }
        mine.push_back({(long)f * 4096 + tile, {s0, s1}});
      }
      if (n == 0) FAIL("xcd %d wg %d: non-empty range without pieces", xcd, wg);
      // walk order is last-first; record in ascending stage order per workgroup
      for (int i = (int)mine.size() - 1; i >= 0; --i) cover[mine[i].first].push_back(mine[i].second);
      *pieces_total += n;
      prev_nonempty = wg;
      prev_stop = pstop;
    }
  }
  // coverage
  long expected = 0;
  for (int f = 0; f < kWF; ++f)
    for (int t = 0; t < tiles; ++t) {
      const int S = x3_item_stages(b, f, t);
      auto it = cover.find((long)f * 4096 + t);
      if (S == 0) {
        if (it != cover.end()) FAIL("absent tile (f %d, tile %d) has work", f, t);
        continue;
      }
      ++expected;
      if (it == cover.end()) FAIL("tile (f %d, tile %d) has no work", f, t);
      int at = 0;
      for (auto& pr : it->second) {
        if (pr.first != at) FAIL("tile (f %d, tile %d): piece starts at %d, expected %d", f, t, pr.first, at);
        at = pr.second;
      }
      if (at != S) FAIL("tile (f %d, tile %d): covered to %d of %d", f, t, at, S);
    }
  if ((long)cover.size() != expected) FAIL("%zu tiles with work, expected %ld", cover.size(), expected);
  return true;
}

int main(int argc, char** argv) {
  std::vector<Shape> shapes;
  for (int i = 1; i + 6 < argc; i += 7)
    shapes.push_back(Shape{atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]), atoi(argv[i + 3]), atoi(argv[i + 4]), atoi(argv[i + 5]), atoi(argv[i + 6])});
  if (shapes.empty()) {
    const int Ms[] = {16, 32, 96, 256, 320, 1024, 4096, 8192}, Ns[] = {32, 256, 512, 1024, 4096}, Ks[] = {64, 96, 256, 512, 2048};
    for (int M : Ms)
      for (int N : Ns)
        for (int K : Ks)
          for (int nw : {1, 4, 32, 38}) {
            shapes.push_back(Shape{M, N, K, 0, 0, 0, nw});
            if (K % 128 == 0) shapes.push_back(Shape{M, N, K, 1, K / 4, 0, nw});
            if (N % 4 == 0 && (N / 4) % 32 == 0) shapes.push_back(Shape{M, N, K, 2, N / 4, kWA - 1, nw});
            if (M % 4 == 0 && (M / 4) % 32 == 0) shapes.push_back(Shape{M, N, K, 3, M / 4, kWA - 1, nw});
          }
  }
  long pieces = 0, parked = 0;
  for (const Shape& sh : shapes)
    if (!check(sh, &pieces, &parked)) return 1;
  printf("ok %zu shapes, %ld pieces, %ld parked\n", shapes.size(), pieces, parked);
  return 0;
}
