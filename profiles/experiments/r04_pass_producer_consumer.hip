// NOT BUILT — kept as the record of an experiment (round 4), with its A/B in profiles/r04_pass_pc_ab.log.
// It was compiled inside csrc/ndt_derivatives.hip (it uses that file's PassParams, pair_update, TilePoint, wave_sync, lsa_store_row32)
// and launched from ndt_enqueue_pass under LSA_PASS_PC=1 with a grid of min(ceil(tiles / 2), 4 x CUs) workgroups of 192 lanes:
//   if (pc) hipLaunchKernelGGL(k_ndt_pass_pc, dim3(g), dim3(kPcBlock), 0, stream, d_source, n_source, P, dense, rec, partials, 0ull);
// Result at BASELINE.json configs[4] size: same pairs, sums equal to 1e-15, 274-290 us per launch against 186-196 us for the
// one-wave form on the same box: the ONE consumer wave of a workgroup carries every record gather's L2 latency (117 k rounds over
// 1 024 consumer waves x ~2 us) and the producers, which need a third of the registers, are held to the consumer's allocation,
// so a CU still holds 12 waves.  DESIGN.md section 8 has the reasoning; the item is closed.
// ---- EXPERIMENT (round 4): producer / consumer waves ------------------------------------------------------
// The pass at 2 M points is neither issue nor bandwidth bound: every wave alternates between a latency chain (source -> transform
// -> probe rows from L2 -> ballot) and a dense FMA chain (pair math), and three waves per SIMD do not cover one with the other
// (VALU ~62 % busy).  Here the two halves live in DIFFERENT waves of a 192-lane workgroup: two PRODUCER waves run phase 0 and the
// dense-row probes of their own tiles and leave each tile's compacted pairs, with the tile's point data, in one of three LDS slots;
// one CONSUMER wave takes the tiles in a fixed order (tile k of producer 0, of producer 1, tile k + 1 ...), gathers full rounds of
// 64 pairs (what is left is carried, as in the one-wave form) and does the record gathers and the pair math into its 28
// accumulators.  Hand-offs are LDS words (produced / consumed tile counts) polled with s_sleep: no s_barrier anywhere, a wave only
// ever waits for the one thing it needs.  The order in which pairs meet is a function of the data alone, so the sums are
// reproducible run to run (they differ from the one-wave form's in the last bits: other lanes add other pairs).
// Selected with LSA_PASS_PC=1 for kind 0 on a dense table through the device-reduced path (lsa_ndt_derivatives_timed): measured,
// DESIGN.md section 8 records what it gave.
namespace {
constexpr int kPcProd = 2;
constexpr int kPcBlock = 64 * (kPcProd + 1);
constexpr int kPcSlots = 3;
constexpr int kPcSlotTerms = 512;                         // behind 16 TilePoints
constexpr int kPcSlotPairs = kPcSlotTerms + kTileSlots * kTermStride * 8;   // 3584
constexpr int kPcSlotBytes = kPcSlotPairs + 1536;         // 378 pair entries, padded: 5120
constexpr int kPcCtl = 0;                                 // ready[2], done[2], npairs[2][3]: 16 words; behind them the consumer's endpos[2][3]
constexpr int kPcEnd = 64;
constexpr int kPcSlotBase = 128;
constexpr int kPcQueue = kPcSlotBase + kPcProd * kPcSlots * kPcSlotBytes;   // consumer queue: < 64 carried + <= 378 fresh
constexpr int kPcJh = kPcQueue + 512 * 4;
constexpr int kPcTl = kPcJh + 72 * 8;
constexpr int kPcLds = kPcTl + 16 * 4;
static_assert(64 * kRedStride * 8 <= kPcProd * kPcSlots * kPcSlotBytes, "the reduction area aliases the tile slots");

__global__ __launch_bounds__(kPcBlock, 3) void k_ndt_pass_pc(const float4 *__restrict__ src, size_t n, PassParams P, const float4 *__restrict__ dense,
                                                            const double *__restrict__ rec, double *__restrict__ partials, unsigned long long key) {
  __shared__ __align__(16) unsigned char lds[kPcLds];
  volatile unsigned *ctl = reinterpret_cast<volatile unsigned *>(lds + kPcCtl);   // [0..1] ready, [2..3] done, [4..9] npairs
  double *jh = reinterpret_cast<double *>(lds + kPcJh);
  float *Tl = reinterpret_cast<float *>(lds + kPcTl);
  if (threadIdx.x < 69) jh[threadIdx.x] = (&P.ja[0][0])[threadIdx.x];
  else if (threadIdx.x < 69 + 12) Tl[threadIdx.x - 69] = P.T[threadIdx.x - 69];
  if (threadIdx.x >= 96 && threadIdx.x < 112) ctl[threadIdx.x - 96] = 0u;
  __syncthreads();   // (the only one: the constants and the zeroed control words)
  const PassScalars Q = pass_scalars(P);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t ntiles = (n + kTilePts - 1) / kTilePts;
  const size_t qstride = (size_t)gridDim.x * kPcProd;
  const int dimx = Q.dim0, dimxy = Q.dim0 * Q.dim1;
  if (wave < kPcProd) {
    // ------------------------------------------------------------------------------------------------ producer
    const int p = wave;
    const int part4 = lane & 3, ppl = lane >> 2;
    PassPre pre;
    pass_pre_init<true>(pre);
    const unsigned probe_lut = pre.probe_lut;
    const size_t tile0 = (size_t)blockIdx.x * kPcProd + p;
    float4 s_next = make_float4(0.f, 0.f, 0.f, 0.f);
    {
      size_t i0 = tile0 * kTilePts + ppl;
      if (ppl < kTilePts && i0 < n) s_next = src[i0];
    }
    unsigned k = 0;
    for (size_t tile = tile0; tile < ntiles; tile += qstride, k++) {
      const unsigned slot = k % kPcSlots;
      unsigned char *sb = lds + kPcSlotBase + (p * kPcSlots + slot) * kPcSlotBytes;
      TilePoint *tp = reinterpret_cast<TilePoint *>(sb);
      double *terms = reinterpret_cast<double *>(sb + kPcSlotTerms);
      unsigned *pairs = reinterpret_cast<unsigned *>(sb + kPcSlotPairs);
      // the slot is free once the consumer has finished the tile that used it last
      while (k >= ctl[2 + p] + kPcSlots) __builtin_amdgcn_s_sleep(2);
      // ---- phase 0 (ndt_pass_body's, one buffer)
      {
        size_t i = tile * kTilePts + ppl;
        float4 s = s_next;
        size_t inext = (tile + qstride) * kTilePts + ppl;
        if (ppl < kTilePts && inext < n) s_next = src[inext];
        {
          int a = part4 < 3 ? part4 : 0;
          asm volatile("" : "+v"(a));
          const int mb = a == 0 ? Q.min_b0 : (a == 1 ? Q.min_b1 : Q.min_b2);
          const int dmax = (a == 0 ? Q.dim0 : (a == 1 ? Q.dim1 : Q.dim2)) - 1;
          const int stride = a == 0 ? 1 : (a == 1 ? dimx : dimxy);
          const float *Ta = Tl + 4 * a;
          const float t = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(Ta[0], s.x), lsa_fmul(Ta[1], s.y)), lsa_fmul(Ta[2], s.z)), Ta[3]);
          const float r = Q.res;
          int h, l;
          if (Q.res_pow2) {
            const float ir = Q.inv_res;
            h = (int)floorf(lsa_fmul(lsa_fadd(t, r), ir)) - mb; l = (int)floorf(lsa_fmul(lsa_fsub(t, r), ir)) - mb;
          } else {
            h = (int)floorf(lsa_fdiv(lsa_fadd(t, r), r)) - mb; l = (int)floorf(lsa_fdiv(lsa_fsub(t, r), r)) - mb;
          }
          h = min(h, dmax);
          const int l0 = l;
          l = max(l, 0);
          const int na = min(h - l + 1, 3);
          const bool ok = ppl < kTilePts && i < n && part4 < 3 && (__float_as_uint(t) & 0x7f800000u) != 0x7f800000u && na > 0;
          unsigned w = ok ? ((((unsigned)na) | ((unsigned)(l - l0) << 2)) << (8 * a)) | (1u << (24 + a)) : 0u;
          int lin = ok ? l * stride : 0;
          const float rr = t - (float)(l0 + mb) * r;
          w += (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0x55, 0xf, 0xf, true) + (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0xaa, 0xf, 0xf, true);
          lin += __builtin_amdgcn_mov_dpp(lin, 0x55, 0xf, 0xf, true) + __builtin_amdgcn_mov_dpp(lin, 0xaa, 0xf, 0xf, true);
          float *tpw = reinterpret_cast<float *>(tp + ppl);
          if (part4 < 3) { tpw[a] = t; tpw[5 + a] = rr; }
          if (part4 == 0) {
            const bool valid = (w >> 24) == 7u;
            tp[ppl].cell0 = valid ? lin : 0;
            tp[ppl].nxyz = valid ? (int)(w & 0xffffffu) : 0;
          }
        }
        double x = s.x, y = s.y, z = s.z;
        const int k0 = part4 * 6;
        double tv[6];
#pragma unroll
        for (int u = 0; u < 6; u++) {
          const int kk = min(k0 + u, kPointTerms - 1);
          tv[u] = x * jh[kk * 3] + y * jh[kk * 3 + 1] + z * jh[kk * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < 6; u++)
          if (k0 + u < kPointTerms) terms[ppl * kTermStride + k0 + u] = tv[u];
      }
      wave_sync();
      // ---- phase A (dense table): row probes -> this tile's pairs
      unsigned total = 0;
      {
        unsigned lut_v = probe_lut;
        asm volatile("" : "+v"(lut_v));
        int r_cell[kRowRounds], r_nx[kRowRounds], r_pl[kRowRounds];
        float r_y[kRowRounds], r_z[kRowRounds];
        float4 e0[kRowRounds], e1[kRowRounds], e2[kRowRounds];
#pragma unroll
        for (int j = 0; j < kRowRounds; j++) {
          const unsigned lut = lut_v >> (16 * j);
          const int pl = lut & 0x3f, dy = (lut >> 8) & 3, dz = (lut >> 10) & 3;
          const bool live = (lut >> 15) & 1;
          int nxyz = tp[pl].nxyz;
          const int iy = dy - ((nxyz >> 10) & 3), iz = dz - ((nxyz >> 18) & 3);
          bool in_box = live && (unsigned)iy < (unsigned)((nxyz >> 8) & 3) && (unsigned)iz < (unsigned)((nxyz >> 16) & 3);
          r_pl[j] = pl;
          r_nx[j] = in_box ? (nxyz & 0xf) : 0;
          r_cell[j] = in_box ? tp[pl].cell0 + __mul24(iy, dimx) + ((iz & 1) ? dimxy : 0) + ((iz & 2) ? 2 * dimxy : 0) : 0;
          r_y[j] = tp[pl].ry - (float)dy * Q.res;
          r_z[j] = tp[pl].rz - (float)dz * Q.res;
        }
#pragma unroll
        for (int j = 0; j < kRowRounds; j++) {
          const float4 *row = dense + r_cell[j];
          e0[j] = row[0]; e1[j] = row[1]; e2[j] = row[2];
        }
#pragma unroll
        for (int j = 0; j < kRowRounds; j++) {
          const float rx = tp[r_pl[j]].rx;
          const float4 ev[3] = {e0[j], e1[j], e2[j]};
#pragma unroll
          for (int i = 0; i < 3; i++) {
            const int ii = i - (r_nx[j] >> 2);
            const float4 en = ii == 0 ? ev[0] : (ii == 1 ? ev[1] : ev[2]);
            float dx = (rx - (float)i * Q.res) - en.x, dy = r_y[j] - en.y, dz = r_z[j] - en.z;
            float d2 = dx * dx + dy * dy + dz * dz;
            int rank = __float_as_int(en.w);
            bool ok = (unsigned)ii < (unsigned)(r_nx[j] & 3) && rank >= 0 && d2 <= Q.r2f_pre;
            const unsigned long long m = __ballot(ok);
            const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            // point | slot | producer | record rank: what the consumer needs to find the point's data and the record
            if (ok) pairs[total + below] = (unsigned)r_pl[j] | (slot << 4) | ((unsigned)p << 6) | ((unsigned)rank << 7);
            total += (unsigned)__popcll(m);
          }
        }
      }
      wave_sync();
      if (lane == 0) {
        ctl[4 + p * kPcSlots + slot] = total;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        ctl[p] = k + 1u;   // the tile is the consumer's
      }
    }
    return;
  }
  // -------------------------------------------------------------------------------------------------- consumer
  double acc[28];
#pragma unroll
  for (int c = 0; c < 28; c++) acc[c] = 0.0;
  unsigned accepted = 0;
  unsigned *cq = reinterpret_cast<unsigned *>(lds + kPcQueue);
  unsigned total = 0;                       // entries in the queue (the carried ones at its front)
  unsigned long long consumed = 0;          // pairs processed so far, as a position in the concatenated sequence
  unsigned appended[kPcProd] = {0u, 0u}, released[kPcProd] = {0u, 0u};
  unsigned long long *endpos = reinterpret_cast<unsigned long long *>(lds + kPcEnd);   // [producer][slot]: where the slot's tile ends in the sequence
  auto run_round = [&](unsigned qi, bool live) {
    const unsigned e = live ? cq[qi] : 0u;
    const unsigned char *sb = lds + kPcSlotBase + (((e >> 6) & 1u) * kPcSlots + ((e >> 4) & 3u)) * kPcSlotBytes;
    const TilePoint *tp = reinterpret_cast<const TilePoint *>(sb) + (e & 15u);
    const double *tm = reinterpret_cast<const double *>(sb + kPcSlotTerms) + (e & 15u) * kTermStride;
    const double *R = reinterpret_cast<const double *>(reinterpret_cast<const char *>(rec) + (e >> 7) * (unsigned)(kRecDoubles * sizeof(double)));
    double pg[8];
#pragma unroll
    for (int c = 0; c < 8; c++) pg[c] = tm[c];
    pair_update<0>(Q, R, tp->tx, tp->ty, tp->tz, pg, tm + 8, acc, live, accepted);
  };
  auto release = [&]() {   // tiles whose every pair has been processed give their slot back
#pragma unroll
    for (int q = 0; q < kPcProd; q++) {
      unsigned r = released[q];
      while (r < appended[q] && endpos[q * kPcSlots + r % kPcSlots] <= consumed) r++;
      if (r != released[q]) { released[q] = r; if (lane == 0) ctl[2 + q] = r; }
    }
  };
  for (unsigned k = 0;; k++) {
    bool any = false;
#pragma unroll
    for (int q = 0; q < kPcProd; q++) {
      const size_t tile = (size_t)blockIdx.x * kPcProd + q + (size_t)k * qstride;
      if (tile >= ntiles) continue;
      any = true;
      // every slot of this producer still holds unprocessed pairs (sparse tiles: < 64 pairs over several tiles): process them now,
      // or the producer could never deliver tile k.  (A function of the pair counts only: the order stays reproducible.)
      if (appended[q] - released[q] == kPcSlots) {
        const bool live = (unsigned)lane < total;
        if (total) run_round((unsigned)lane, live);
        consumed += total;
        total = 0;
        wave_sync();
        release();
      }
      while (ctl[q] <= k) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const unsigned slot = k % kPcSlots;
      const unsigned np = ctl[4 + q * kPcSlots + slot];
      const unsigned *pairs = reinterpret_cast<const unsigned *>(lds + kPcSlotBase + (q * kPcSlots + slot) * kPcSlotBytes + kPcSlotPairs);
      for (unsigned i = lane; i < np; i += 64) cq[total + i] = pairs[i];
      total += np;
      if (lane == 0) endpos[q * kPcSlots + slot] = consumed + total;
      appended[q] = k + 1u;
      wave_sync();
      const unsigned full = total >> 6;
      for (unsigned r = 0; r < full; r++) run_round(r * 64 + lane, true);
      const unsigned left = total & 63u;
      if (full != 0 && left != 0) {
        unsigned e = (unsigned)lane < left ? cq[full * 64 + lane] : 0u;
        wave_sync();
        if ((unsigned)lane < left) cq[lane] = e;
      }
      consumed += (unsigned long long)full * 64ull;
      total = left;
      wave_sync();
      release();
    }
    if (!any) break;
  }
  if (total) run_round((unsigned)lane, (unsigned)lane < total);
  // ---- the consumer's 64 x 28 accumulators -> one row (fixed order), with the pair and point counts
  wave_sync();
  double *red = reinterpret_cast<double *>(lds + kPcSlotBase);
#pragma unroll
  for (int c = 0; c < 28; c++) red[lane * kRedStride + c] = acc[c];
  wave_sync();
  double v = 0.0;
  if (lane < 28) {
    double c0 = red[lane], c1 = red[kRedStride + lane], c2 = red[2 * kRedStride + lane], c3 = red[3 * kRedStride + lane];
#pragma unroll 4
    for (int c = 4; c < 64; c += 4) {
      c0 += red[c * kRedStride + lane]; c1 += red[(c + 1) * kRedStride + lane]; c2 += red[(c + 2) * kRedStride + lane]; c3 += red[(c + 3) * kRedStride + lane];
    }
    v = (c0 + c1) + (c2 + c3);
  } else if (lane == 28) {
    v = (double)accepted;
  } else if (lane == 29) {
    unsigned long long npts = 0;
    for (int q = 0; q < kPcProd; q++) {
      const size_t t0 = (size_t)blockIdx.x * kPcProd + q;
      if (t0 < ntiles) {
        const size_t last = ntiles - 1 - t0;
        npts += (unsigned long long)(last / qstride + 1) * kTilePts;
        if (last % qstride == 0) npts -= (unsigned long long)(ntiles * kTilePts - n);
      }
    }
    v = (double)npts;
  }
  if (lane < kOutDoubles) lsa_store_row32(partials + (size_t)blockIdx.x * kOutDoubles, lane, v, key);
}
}  // namespace

