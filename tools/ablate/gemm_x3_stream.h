// The persistent "stream-K inside an XCD" variant of the split-precision GEMM (rounds 2 - 4 of the build; csrc/gemm_x3.h until
// round 5).  Measured slower than the 256 x 128 two-workgroups-per-CU kernel on every DCGAN / DenseNet shape (DESIGN section 3:
// 4096x1024x256 304 -> 351 -> 302 us, ...), selected by no default path since round 3 and launched zero times in the round-4
// traces: moved out of the product library.  Kept as an ablation artefact for tools/ablate/x3_phase.hip
// (phase timing, one-tile against stream) and tools/ablate/stream_plan_check.hip (host check of the work partition).
// Include AFTER csrc/gemm_x3.h, with -DX3_STREAM_TOOL (the BgArgs fields sk_* exist only then).
#pragma once
#ifndef X3_STREAM_TOOL
#error "gemm_x3_stream.h needs -DX3_STREAM_TOOL (BgArgs::sk_* fields)"
#endif
namespace {
// ---- the same GEMM as ONE persistent workgroup per compute unit ------------------------------------
// Measured on the Winograd-domain shapes (tools/ablate/x3_phase.hip, cycles per 256 x 256 tile): the K loop itself
// runs at 91 % of the matrix-pipe rate (3380 cycles per stage of 96 MFMAs), but a workgroup that owns one tile
// spends 9 k cycles waiting for its first operands, 12 k writing C in a burst that every compute unit issues at the
// same moment, and the next workgroup reaches the compute unit ~20 k cycles later -- against 54 k cycles of MFMA work
// at K = 256.  And a frequency-major grid leaves each XCD with 18 .. 288 tiles for 32 compute units (1.1 .. 9
// rounds, the last one partly empty).  Here every XCD's queue -- its frequencies' tiles, each a run of K stages --
// is cut into sk_nw equal contiguous ranges of stages, one per workgroup ("stream-K" within an XCD):
//  * the operand stream is continuous across tile boundaries: the loads of the next tile's first stages are in
//    flight while the current tile finishes and is written out;
//  * a range boundary may fall inside a tile.  A workgroup walks its range BACKWARDS, so the head part of a shared
//    tile is the first thing its workgroup computes -- parked in the workspace and flagged -- and the tail part the
//    last thing the next workgroup computes; that one adds the parked sums (fixed order: deterministic) and writes
//    C.  A workgroup only ever waits for lower-numbered workgroups, which were dispatched before it and park their
//    sums before waiting for anything themselves: no deadlock, whatever is resident;
//  * boundaries inside tiles also spread the C writes of the compute units over time.
constexpr int X3_SK_MAXW = 40;
constexpr size_t X3_SK_SLOT = (size_t)X3_BM * X3_BN;
inline size_t x3_stream_ws_floats(int grid) { return (size_t)grid * X3_SK_SLOT + 2 * (size_t)grid + 4; }
__host__ __device__ __forceinline__ unsigned x3_pos(int slot, int tile, int stage) {
  return ((unsigned)slot << 28) | ((unsigned)tile << 16) | (unsigned)stage;
}
// tile -> (row tile, column tile) with the precomputed reciprocal (exact for tile < 4096)
__host__ __device__ __forceinline__ int x3_tile_row(const BgArgs& a, int tile) {
  return (int)(((unsigned long long)(unsigned)tile * a.sk_inv_tn) >> 32);   // < 2^12 * 2^32
}
// K stages (of 16) of tile `tile` at frequency f: 0 = structurally absent (strided dgrad / wgrad)
__host__ __device__ __forceinline__ int x3_item_stages(const BgArgs& a, int f, int tile) {
  const int full = a.K / X3_SK;
  if (a.seg_mode == 0) return full;
  if (a.seg_mode == 1) {
    int w = 0;
    for (int c = 0; c < 4; ++c) w += s2_present(c, f, a.seg_skip) ? 1 : 0;
    return w * (a.seg_len / X3_SK);
  }
  const int tm = x3_tile_row(a, tile), tn = tile - tm * a.tiles_n;
  const int mask = a.sk_cmask[a.seg_mode == 2 ? tn : tm];
  bool any = false;
  for (int c = 0; c < 4; ++c) any = any || (((mask >> c) & 1) && s2_present(c, f, a.seg_skip));
  return any ? full : 0;
}
__host__ __device__ __forceinline__ void x3_slot_info(const BgArgs& a, int xcd, int slot, int& f, int& t0, int& t1) {
  const int code = (unsigned char)a.fmap[xcd][slot];
  const int tiles = a.tiles_m * a.tiles_n, half = (tiles + 1) >> 1;
  f = code & 63;
  t0 = (code & 128) ? half : 0;
  t1 = (code & 64) ? half : tiles;
}

// A workgroup's range [pstart, pstop) of its XCD's queue, walked BACKWARDS one piece at a time (device: the piece
// table of the stream kernel; host: tests/test_stream_plan_cpu.py through tools/stream_plan_check.hip).
struct X3Walk {
  unsigned end;            // everything at or after this position has been handed out
  int slot, f, t0, t1;     // the slot `end` lies in: its frequency and tile range
};
__host__ __device__ __forceinline__ void x3_walk_begin(const BgArgs& a, int xcd, unsigned pstop, X3Walk& w) {
  w.end = pstop;
  w.slot = (int)(pstop >> 28);
  x3_slot_info(a, xcd, w.slot, w.f, w.t0, w.t1);
}
// next piece (stages [s0, s1) of the S stages of (f, tile); lo = position of the tile's stage 0); false: range done
__host__ __device__ __forceinline__ bool x3_walk_next(const BgArgs& a, int xcd, unsigned pstart, X3Walk& w, int& f, int& tile,
                                                      int& s0, int& s1, int& S, unsigned& lo) {
  while (w.end > pstart) {
    tile = (int)((w.end >> 16) & 0xfffu);
    int stage = (int)(w.end & 0xffffu);
    if (stage == 0) {
      if (tile > w.t0) {
        tile -= 1;
      } else {
        --w.slot;
        x3_slot_info(a, xcd, w.slot, w.f, w.t0, w.t1);
        tile = w.t1 - 1;
      }
      w.end = x3_pos(w.slot, tile, 0);
    }
    S = x3_item_stages(a, w.f, tile);
    if (stage == 0) {
      if (S == 0) continue;
      stage = S;
    }
    lo = x3_pos(w.slot, tile, 0);
    s0 = pstart > lo ? (int)(pstart & 0xffffu) : 0;
    s1 = stage;
    f = w.f;
    w.end = lo + (unsigned)s0;
    return true;
  }
  return false;
}

// One contiguous piece of one tile's contraction as the kernel keeps it in LDS (built once per workgroup): stages
// [s0, s1) of the S stages of (f, tile), lo = queue position of the tile's stage 0, and the operand stream of the
// piece (base of the frequency's A / B planes, first row / column block of the tile, stage -> k block map).
struct X3Piece {
  const u16* opA;
  const u16* opB;
  int oA, oB;              // first row block (of 32) / column block (of 16, TL) of the tile in A / B
  int steps0, kb0, kb1;    // stage j -> k block: j < steps0 ? kb0 + j : kb1 + j (two class runs: strided forward)
  int s0, s1, S;
  int f, tile;
  unsigned lo;
  int pad;
};
static_assert(sizeof(X3Piece) == 64, "X3Piece is read as four 16-byte words");
constexpr int X3_SK_MAXPIECES = 160;
constexpr size_t X3_SK_LDS = X3_LDS + 512 + (size_t)X3_SK_MAXPIECES * sizeof(X3Piece);

template <bool TL>
__global__ __launch_bounds__(X3_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_bgemm_x3_stream_kernel(BgArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  const int x = blockIdx.x, xcd = x & 7, wg = x >> 3;
  const unsigned pstart = a.sk_bound[xcd][wg], pstop = a.sk_bound[xcd][wg + 1];
  if (pstart >= pstop) return;
  X3_STAMP(0);
  int stamp_i = 2;
  (void)stamp_i;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, g = lane >> 5;

  // ---- this workgroup's pieces, last first (every thread walks the queue; thread 0 writes the table).  The first
  // piece alone is enough to start the operand stream (a piece has >= 4 stages): the rest of the table is built
  // while the first three stages are in flight.
  X3Piece* pieces = reinterpret_cast<X3Piece*>(smem3 + X3_LDS + 512);
  int npieces = 0;
  X3Walk wk;
  x3_walk_begin(a, xcd, pstop, wk);
  auto walk = [&](int limit) {
    while (npieces < limit) {
      int f, tile, s0, stage, S;
      unsigned lo;
      if (!x3_walk_next(a, xcd, pstart, wk, f, tile, s0, stage, S, lo)) break;
      X3Piece pc;
      const int tm = x3_tile_row(a, tile), tn = tile - tm * a.tiles_n;
      pc.opA = a.Ap + (long)f * a.sAp;
      pc.opB = a.Bp + (long)f * a.sBp;
      pc.oA = TL ? (tm * X3_BM) >> 4 : (tm * X3_BM) >> 5;
      pc.oB = TL ? (tn * X3_BN) >> 4 : (tn * X3_BN) >> 5;
      int lo0 = 0, len0 = a.K, lo1 = 0;
      if (a.seg_mode == 1) {
        int cc = 0, nrun = 0;
        len0 = 0;
        while (cc < 4) {
          if (!s2_present(cc, f, a.seg_skip)) {
            ++cc;
            continue;
          }
          int e = cc + 1;
          while (e < 4 && s2_present(e, f, a.seg_skip)) ++e;
          if (nrun == 0) {
            lo0 = cc * a.seg_len;
            len0 = (e - cc) * a.seg_len;
          } else {
            lo1 = cc * a.seg_len;
          }
          ++nrun;
          cc = e;
        }
      }
      pc.steps0 = len0 / X3_SK;
      pc.kb0 = lo0 / X3_SK;
      pc.kb1 = lo1 / X3_SK - pc.steps0;
      pc.s0 = s0; pc.s1 = stage; pc.S = S;
      pc.f = f; pc.tile = tile; pc.lo = lo; pc.pad = 0;
      if (tid == 0) pieces[npieces] = pc;
      ++npieces;
    }
  };
  walk(1);
  if (npieces == 0) return;   // (the planner never hands out such a range)
  __syncthreads();
  // a piece's fields as scalars (all lanes read the same words)
  auto piece_words = [&](int idx, int w) -> u32x4 {
    const u32x4 v = reinterpret_cast<const u32x4*>(pieces + idx)[w];
    u32x4 s;
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = (unsigned)__builtin_amdgcn_readfirstlane((int)v[e]);
    return s;
  };

  // ---- operand stream (this wave: 12 chunks per stage of A (waves 0,1) or B (waves 2,3)) --------------------
  const bool isA = wave < 2;
  const int half = wave & 1;
  const long plane = isA ? a.pA : a.pB;
  const int rbmax = (isA ? a.rbA : a.rbB) - 1, cbn = isA ? a.cbA : a.cbB;
  const unsigned voff = TL ? (unsigned)(lane >> 5) * 1024u + (unsigned)(lane & 31) * 16u : (unsigned)lane * 16u;
  const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem3;
  struct Cur {
    const u16* base[X3_PER_WAVE];   // per chunk stream: plane + row / column block of the tile, k block 0
    int steps0, kb0, kb1, j, s1, idx;
  };
  auto setup = [&](Cur& c, int idx) {
    const u32x4 w0 = piece_words(idx, 0), w1 = piece_words(idx, 1), w2 = piece_words(idx, 2);
    const unsigned long long pa = ((unsigned long long)w0[1] << 32) | w0[0], pb = ((unsigned long long)w0[3] << 32) | w0[2];
    const u16* opb = reinterpret_cast<const u16*>(isA ? pa : pb);
    const int o0 = (int)(isA ? w1[0] : w1[1]);
#pragma unroll
    for (int i = 0; i < X3_PER_WAVE; ++i) {
      const int li = half * X3_PER_WAVE + i;
      const int piece = li >> 3, rg = li & 7;
      if (TL) {
        int cb = o0 + 2 * rg;
        if (cb > ((cbn - 1) & ~1)) cb = (cbn - 1) & ~1;
        c.base[i] = opb + piece * plane + ((long)cb << 9);
      } else {
        int rb = o0 + rg;
        if (rb > rbmax) rb = rbmax;
        c.base[i] = opb + piece * plane + (((long)rb * a.kblocks) << 9);
      }
    }
    c.steps0 = (int)w1[2];
    c.kb0 = (int)w1[3];
    c.kb1 = (int)w2[0];
    c.j = (int)w2[1];
    c.s1 = (int)w2[2];
    c.idx = idx;
  };
  auto issue = [&](const Cur& c, int buf, int i0, int n) {
    const long kb = c.j < c.steps0 ? c.kb0 + c.j : c.kb1 + c.j;
    const long koff = TL ? (((kb >> 1) * cbn) << 9) + ((kb & 1) << 8) : kb << 9;
#pragma unroll
    for (int i = i0; i < i0 + n; ++i) {
      const int li = half * X3_PER_WAVE + i;
      const int piece = li >> 3, rg = li & 7;
      const u16* src = c.base[i] + koff;
      const unsigned dst = lds_base + buf * X3_STAGE + (isA ? 0 : X3_NP * X3_TA) + piece * X3_TA + rg * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst) : "memory");
    }
  };
  // next stage of the stream; past the end of the range the last stage is fetched again (into a free buffer, never
  // read): the number of loads in flight stays the same for every stage
  auto advance = [&](Cur& c) {
    if (++c.j == c.s1) {
      if (c.idx + 1 < npieces) setup(c, c.idx + 1);
      else c.j = c.s1 - 1;
    }
  };

  f32x16 acc[X3_MT][X3_NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < X3_MT; ++i)
#pragma unroll
      for (int j = 0; j < X3_NT; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  };
  const int sw = (r >> 3) & 1;
  const int g4 = lane >> 4, l16 = lane & 15;
  const int ftl = (g4 & 1) * 512 + (8 * (g4 >> 1) + (l16 >> 2)) * 32 + ((((l16 >> 1) & 1) ^ (g4 >> 1)) * 16) + (l16 & 1) * 8;
  const int fa = TL ? wm * 4096 + ftl : (wm * X3_MT * 32 + r) * 32 + 16 * (g ^ sw);
  const int fb = X3_NP * X3_TA + (TL ? wn * 4096 + ftl : (wn * X3_NT * 32 + r) * 32 + 16 * (g ^ sw));
  auto read_frag = [&](const unsigned char* p) -> x3frag_t {
    if (TL) {
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 128));
      return __builtin_bit_cast(x3frag_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    } else {
      return *reinterpret_cast<const x3frag_t*>(p);
    }
  };
  // one stage (see the one-tile kernel): the stage after this one has landed (vmcnt + barrier; `landed`: known
  // without waiting), the buffer this stage was read from is refilled three stages ahead, 96 MFMAs on F with the
  // 24 fragment reads of the next stage (into G) in between
  auto stage = [&](Cur& c, int bufn, const X3Frags& F, X3Frags& G, bool landed) {
    if (!landed) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(X3_PER_WAVE) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (fragment reads returned before their buffer is refilled)
    __builtin_amdgcn_s_barrier();
    const int rbuf = bufn == 0 ? X3_NSTAGE - 1 : bufn - 1;
    const unsigned char* pa = smem3 + bufn * X3_STAGE + fa;
    const unsigned char* pb = smem3 + bufn * X3_STAGE + fb;
    constexpr int NR = 8 * X3_NP;
    x3_static_for<X3_NTERM>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int l0 = t * X3_PER_WAVE / X3_NTERM, l1 = (t + 1) * X3_PER_WAVE / X3_NTERM;
      constexpr int q0 = t * NR / X3_NTERM, q1 = (t + 1) * NR / X3_NTERM;
      issue(c, rbuf, l0, l1 - l0);
#pragma unroll
      for (int q = q0; q < q1; ++q) {
        const int qq = q % (4 * X3_NP), tt = qq / X3_NP, p = qq % X3_NP;
        if (q < 4 * X3_NP) G.a[tt][p] = read_frag(pa + p * X3_TA + tt * 1024);
        else G.b[tt][p] = read_frag(pb + p * X3_TB + tt * 1024);
      }
#pragma unroll
      for (int i = 0; i < X3_MT; ++i)
#pragma unroll
        for (int j = 0; j < X3_NT; ++j) acc[i][j] = X3_MFMA(F.a[i][X3_PA[t]], F.b[j][X3_PB[t]], acc[i][j]);
      x3_sched_group<TL ? 2 : 1, q1 - q0>();
    });
    advance(c);
  };

  // ---- what happens to a finished piece ------------------------------------------------------------------
  // Sums of a tile whose contraction is shared are chained through the workspace: a workgroup adds the parked
  // sums of the (nearest non-empty) workgroup below it to its own on the way out -- into C when the tile's
  // last stage is here, else into its own slot.  The accumulators themselves are only ever read here.
  unsigned long long* flags = reinterpret_cast<unsigned long long*>(a.sk_partial + (size_t)gridDim.x * X3_SK_SLOT);
  auto finish = [&](int idx) {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const u32x4 w2 = piece_words(idx, 2), w3 = piece_words(idx, 3);
    const int p_s0 = (int)w2[1], p_s1 = (int)w2[2], p_S = (int)w2[3], p_f = (int)w3[0], p_tile = (int)w3[1];
    const bool has_prev = p_s0 > 0;
    const int my_xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15;   // HW_REG_XCC_ID
    bool poison = false;
    // Parked sums stay inside one XCD (workgroup x runs on XCD x % 8 -- tools/ablate/xcc_probe.hip -- and shares
    // tiles only with x - 8 and x + 8), whose L2 both workgroups write through and read from: no cache writeback is
    // needed (a device-scope release fence writes back the whole L2, C tiles and all: 50 - 90 k cycles per parked
    // tile; device-scope loads / stores: slower still).  Producer: plain stores, vmcnt(0) in every thread, barrier,
    // flag = epoch + its XCD.  Consumer: flag seen (and the XCD is its own, else the tile is poisoned: loud, not
    // subtly wrong), barrier, L1 invalidate, plain loads.
    const f32x4v* Pp = nullptr;
    if (has_prev) {
      int k = wg - 1;
      while (k > 0 && a.sk_bound[xcd][k] >= a.sk_bound[xcd][k + 1]) --k;
      const int xk = xcd + 8 * k;
      if (tid == 0) {
        int spins = 0;
        unsigned long long fl;
        while (((fl = __hip_atomic_load(&flags[xk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~15ull) != a.sk_epoch) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1 << 21)) break;   // ~ 1 s: never on a healthy device; a poisoned tile instead of a hung one
        }
        __hip_atomic_store(&flags[xk], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (fl != a.sk_epoch + (unsigned long long)my_xcc) poison = true;
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      Pp = reinterpret_cast<const f32x4v*>(a.sk_partial + (size_t)xk * X3_SK_SLOT) + (size_t)wave * 4096 + lane;
    }
    if (p_s1 < p_S) {
      // the tile's contraction continues in the next workgroup: park the sums (register order, 16 bytes per lane)
      f32x4v* P = reinterpret_cast<f32x4v*>(a.sk_partial + (size_t)x * X3_SK_SLOT) + (size_t)wave * 4096 + lane;
#pragma unroll
      for (int i = 0; i < X3_MT; ++i)
#pragma unroll
        for (int j = 0; j < X3_NT; ++j)
#pragma unroll
          for (int qh = 0; qh < 4; ++qh) {
            const int e = ((i * X3_NT + j) * 4 + qh) * 64;
            f32x4v v = {acc[i][j][4 * qh], acc[i][j][4 * qh + 1], acc[i][j][4 * qh + 2], acc[i][j][4 * qh + 3]};
            if (has_prev) v += Pp[e];
            if (poison) v[0] = __builtin_nanf("");
            P[e] = v;
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        __hip_atomic_store(&flags[x], a.sk_epoch + (unsigned long long)my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    const int tm = x3_tile_row(a, p_tile), tn = p_tile - tm * a.tiles_n;
    const int m0 = tm * X3_BM, n0 = tn * X3_BN;
    float* C = a.C + (long)p_f * a.sC;
    const float es = a.hdrA ? x3_out_scale(a, p_f) : 1.f;   // a power of two: exact
    if (m0 + X3_BM <= a.M && n0 + X3_BN <= a.N) {
      const long ld = a.ldc;
      float* row = C + (long)(m0 + wm * X3_MT * 32 + 4 * g) * ld + (n0 + wn * X3_NT * 32 + r);
      if (!has_prev) {
        // the common case, kept as lean as the one-tile kernel's write-out: one running row pointer, immediate offsets
#pragma unroll
        for (int i = 0; i < X3_MT; ++i) {
#pragma unroll
          for (int qh = 0; qh < 4; ++qh) {
            // (a fence for the scheduler: left alone it copies the whole accumulator tile to vector registers first,
            // spills a third of the copies to scratch and then waits on the reloads with the stores in flight)
            __builtin_amdgcn_sched_barrier(0);
            float* q = row + (long)(i * 32 + 8 * qh) * ld;
#pragma unroll
            for (int ql = 0; ql < 4; ++ql) {
#pragma unroll
              for (int j = 0; j < X3_NT; ++j) q[j * 32] = acc[i][j][4 * qh + ql] * es;
              q += ld;
            }
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < X3_MT; ++i) {
#pragma unroll
          for (int qh = 0; qh < 4; ++qh) {
            __builtin_amdgcn_sched_barrier(0);
            f32x4v pv[X3_NT];
#pragma unroll
            for (int j = 0; j < X3_NT; ++j) pv[j] = Pp[((i * X3_NT + j) * 4 + qh) * 64];
            if (poison) pv[0][0] = __builtin_nanf("");
            float* q = row + (long)(i * 32 + 8 * qh) * ld;
#pragma unroll
            for (int ql = 0; ql < 4; ++ql) {
#pragma unroll
              for (int j = 0; j < X3_NT; ++j) q[j * 32] = (acc[i][j][4 * qh + ql] + pv[j][ql]) * es;
              q += ld;
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < X3_MT; ++i)
#pragma unroll
        for (int j = 0; j < X3_NT; ++j)
#pragma unroll
          for (int qh = 0; qh < 4; ++qh) {
            f32x4v pv = {0.f, 0.f, 0.f, 0.f};
            if (has_prev) pv = Pp[((i * X3_NT + j) * 4 + qh) * 64];
#pragma unroll
            for (int ql = 0; ql < 4; ++ql) {
              const int m = m0 + (wm * X3_MT + i) * 32 + ql + 8 * qh + 4 * g;
              const int n = n0 + (wn * X3_NT + j) * 32 + r;
              if (m < a.M && n < a.N) C[(long)m * a.ldc + n] = (acc[i][j][4 * qh + ql] + pv[ql]) * es;
            }
          }
      X3_WAIT_VM(0);   // (a wave may have issued fewer than 64 stores here)
    }
  };

  // ---- the stream -------------------------------------------------------------------------------------------
  Cur ic;
  setup(ic, 0);
  issue(ic, 0, 0, X3_PER_WAVE);
  ++ic.j;
  issue(ic, 1, 0, X3_PER_WAVE);
  ++ic.j;
  issue(ic, 2, 0, X3_PER_WAVE);
  walk(X3_SK_MAXPIECES);
  npieces = __builtin_amdgcn_readfirstlane(npieces);
  __syncthreads();
  advance(ic);
  X3_WAIT_VM(2 * X3_PER_WAVE);
  __builtin_amdgcn_s_barrier();
  X3Frags F0, F1;
  auto load_frags = [&](X3Frags& F, int buf) {
    const unsigned char* pa = smem3 + buf * X3_STAGE + fa;
    const unsigned char* pb = smem3 + buf * X3_STAGE + fb;
#pragma unroll
    for (int t = 0; t < X3_MT; ++t)
#pragma unroll
      for (int p = 0; p < X3_NP; ++p) F.a[t][p] = read_frag(pa + p * X3_TA + t * 1024);
#pragma unroll
    for (int t = 0; t < X3_NT; ++t)
#pragma unroll
      for (int p = 0; p < X3_NP; ++p) F.b[t][p] = read_frag(pb + p * X3_TB + t * 1024);
  };
  int bufn = 1;           // buffer of the stage after the one about to be computed
  bool landed = false;
  for (int ci = 0; ci < npieces; ++ci) {
    // one piece: the accumulators start from zero and the first fragments come out of LDS here (kept across the
    // write-out of the previous piece they would cost 96 registers; their buffer is refilled only after the next
    // barrier), then pairs of stages
    const u32x4 w2 = piece_words(ci, 2);
    zero_acc();
    load_frags(F0, bufn == 0 ? X3_NSTAGE - 1 : bufn - 1);
    X3_STAMP(stamp_i);
    for (int n = ((int)w2[2] - (int)w2[1]) >> 1; n > 0; --n) {
      stage(ic, bufn, F0, F1, landed);
      bufn = bufn == X3_NSTAGE - 1 ? 0 : bufn + 1;
      stage(ic, bufn, F1, F0, landed);
      bufn = bufn == X3_NSTAGE - 1 ? 0 : bufn + 1;
      landed = false;
    }
    X3_STAMP(stamp_i + 1);
    finish(ci);
    X3_STAMP(stamp_i + 2);
    // Loads and stores retire in issue order: once all but 63 of the operations issued so far have retired, the
    // two stages that were in flight before the >= 64 stores of the write-out are in LDS, and the stores drain
    // under the next two stages.
    X3_WAIT_VM(63);
    X3_STAMP(stamp_i + 3);
#ifdef X3_TIMING
    stamp_i += 4;
#endif
    landed = true;
  }
  X3_WAIT_VM(0);   // the refetched last stages are still landing in LDS
#ifdef X3_TIMING
  X3_STAMP(1);
  __syncthreads();
  if (threadIdx.x < 64) a.dbg[(long)blockIdx.x * 64 + threadIdx.x] = reinterpret_cast<unsigned long long*>(smem3 + X3_LDS)[threadIdx.x];
#endif
}

// Host: cut every XCD's queue (BgArgs::fmap, set by build_fmap) into nw ranges of equal K-stage counts.  False when
// the shape does not fit the position encoding (the caller then launches the one-tile kernel).
inline bool x3_plan_stream(BgArgs& b, int nw, int stagger = 0) {
  const int tiles = b.tiles_m * b.tiles_n;
  if (nw < 1 || nw > X3_SK_MAXW || tiles >= 4096 || b.K / X3_SK >= 65536 || b.K % X3_BK) return false;
  if (b.seg_mode == 1 && b.seg_len % X3_BK) return false;
  b.sk_inv_tn = (0x100000000ull + (unsigned)b.tiles_n - 1) / (unsigned)b.tiles_n;
  memset(b.sk_cmask, 0, sizeof(b.sk_cmask));
  if (b.seg_mode == 2 || b.seg_mode == 3) {
    const int nt = b.seg_mode == 2 ? b.tiles_n : b.tiles_m, ext = b.seg_mode == 2 ? b.N : b.M;
    if (nt > 64) return false;
    for (int t = 0; t < nt; ++t) {
      const int lo = t * 256;
      int hi = lo + 255;
      if (hi >= ext) hi = ext - 1;
      for (int c = lo / b.seg_len; c <= hi / b.seg_len && c < 4; ++c) b.sk_cmask[t] |= (unsigned char)(1 << c);
    }
  }
  struct Item {
    int slot, tile, S;
  };
  static thread_local Item items[8 * 4096];
  for (int xcd = 0; xcd < 8; ++xcd) {
    int n = 0, last_slot = 0, last_t1 = 0;
    long U = 0;
    for (int slot = 0; slot < 8 && b.fmap[xcd][slot] != -1; ++slot) {
      int f, t0, t1;
      x3_slot_info(b, xcd, slot, f, t0, t1);
      for (int t = t0; t < t1; ++t) {
        const int S = x3_item_stages(b, f, t);
        if ((S & 1) || (S > 0 && S < 4)) return false;   // (the kernel starts three stages deep inside its first piece)
        items[n++] = Item{slot, t, S};
        U += S;
      }
      last_slot = slot;
      last_t1 = t1;
    }
    unsigned* bd = b.sk_bound[xcd];
    if (n == 0 || U == 0) {
      for (int c = 0; c <= X3_SK_MAXW; ++c) bd[c] = 0;
      continue;
    }
    const unsigned stop = x3_pos(last_slot, last_t1, 0);
    bd[0] = x3_pos(items[0].slot, items[0].tile, 0);
    int idx = 0;
    long pre = 0;   // stages before items[idx]
    for (int c = 1; c < nw; ++c) {
      long target = U * c / nw;
      // stagger: when whole tiles would line up on the cuts, every compute unit writes its C tile at the same
      // moment; shifting cut c by (c mod 4) quarter tiles spreads the writes (work per workgroup: +- a quarter tile)
      if (stagger && idx < n) target += (long)(c & 3) * (items[idx].S / 4);
      if (target > U) target = U;
      target &= ~1L;
      while (idx < n && pre + items[idx].S <= target) pre += items[idx++].S;
      unsigned pos;
      if (idx >= n) {
        pos = stop;
      } else {
        long off = target - pre;
        // no slivers: a cut closer than four stages to a tile boundary moves onto it
        if (off < 4) {
          off = 0;
        } else if (items[idx].S - off < 4) {
          pre += items[idx++].S;
          off = 0;
        }
        // (a cut in front of structurally absent tiles moves behind them: a range never holds absent tiles only)
        if (off == 0)
          while (idx < n && items[idx].S == 0) ++idx;
        pos = idx >= n ? stop : x3_pos(items[idx].slot, items[idx].tile, (int)off);
      }
      // (two cuts in one tile: at least four stages apart, else the range between them is empty)
      if (pos < bd[c - 1] || ((pos >> 16) == (bd[c - 1] >> 16) && pos - bd[c - 1] < 4)) pos = bd[c - 1];
      bd[c] = pos;
    }
    for (int c = nw; c <= X3_SK_MAXW; ++c) bd[c] = stop;
    // pieces per workgroup must fit the kernel's table
    for (int c = 0; c < nw; ++c) {
      if (bd[c] >= bd[c + 1]) continue;
      X3Walk w;
      x3_walk_begin(b, xcd, bd[c + 1], w);
      int cnt = 0, f, tile, s0, s1, S;
      unsigned lo;
      while (x3_walk_next(b, xcd, bd[c], w, f, tile, s0, s1, S, lo))
        if (++cnt > X3_SK_MAXPIECES) return false;
    }
  }
  b.sk_nw = nw;
  return true;
}

}  // namespace
