// voxel_index.h — internal: sparse voxel index shared by the VoxelFilter, the NDT target grid and the ICP neighbour grid.
//
// Data layout in HBM (all device-resident, grow-only):
//   bits[W]        u32   occupancy bitmap over the dense cell box, cell = i + j*dx + k*dx*dy (x fastest),
//                        W = ceil(ncells / 32).  At KITTI size this is 20 KB (NDT, 1 m cells) to ~1 MB (0.3 m filter
//                        over a 250 m map): L2-resident, unlike a dense record grid.
//   prefix[W]      u32   exclusive prefix of popcount(bits): rank(cell) = prefix[w] + popc(bits[w] & below(b)).
//                        Ranks enumerate occupied cells in ASCENDING linear index, which is exactly the output order
//                        PCL's sort produces, so no sort is needed anywhere.
//   cell_of_rank[V] i32  inverse map (cell linear index per occupied rank).
//   acc[V][NS]     i64   per-cell fixed-point sums (order-independent => bitwise reproducible),
//   cnt[V]         i32   per-cell point counts.
//   dparams        the grid definition itself (CellMap, scales, W, V, status) lives in device memory: the bounding-box
//                  kernel derives it on the device, so the whole pipeline is enqueued without a host round trip and the
//                  host reads it back once, together with the result count.
#ifndef LSA_VOXEL_INDEX_H_
#define LSA_VOXEL_INDEX_H_

#include "lsa_common.h"

struct CellMap {
  int mode;  // 0: PCL VoxelGrid (floor(x * inv_leaf) - (float)min_b); 1: NDT VoxelGrid (floor(x / leaf) - min_b)
  float leaf[3];
  float inv[3];
  int min_b[3];
  int dim[3];
  long long ncells;
};

struct FixScales {
  double s1, inv_s1;  // first moments  (metres)
  double s2, inv_s2;  // second moments (metres^2)
  double sw, inv_sw;  // intensity / static weight
  int fast_round;     // every scaled term stays below 2^51: round-to-nearest-even by adding 1.5 * 2^52 (vi_round_ll), same bits as rint
  int pad;
};

enum { VI_OK = 0, VI_CAPACITY = 1, VI_RANGE = 2, VI_EMPTY = 3, VI_SPEC_MISS = 4 };

struct DevParams {
  CellMap map;
  FixScales fs;
  unsigned long long W;         // bitmap words in use
  unsigned long long W_cap;     // bitmap words allocated (input)
  unsigned long long n_finite;
  unsigned long long n_cap;     // largest point count the fixed-point scales `fs` are safe for
  int spec_miss;                // a speculative build met a finite point outside the reused cell box (result invalid)
  int expand;                   // input: define the cell box with a margin around the bounding box (filter mode only)
  unsigned nsb;                 // scan blocks in use
  unsigned V;                   // occupied cells (written by the scan)
  int status;                   // VI_*
  int req_mode;                 // requested rule set (input)
  float req_leaf[3];            // requested leaf (input)
  float mn[3], mx[3];           // bounding box of the finite points
};

// What the host already knows of the grid definition when a build REUSES the previous one (speculative builds of the frame
// filter): handed to the kernels by value, so that none of them starts with a dependent (and, right after a kernel boundary,
// cold: ~2 us) load of *dparams before it can touch the points.  valid == 0: read the definition from device memory.
struct KnownParams {
  int valid;
  unsigned nsb;
  unsigned long long W;
  CellMap map;
  FixScales fs;
};

// rows the host is still packing into device-visible pinned memory when the build is enqueued (k_mark<.., PULL>, voxel_index.hip)
constexpr unsigned kPullFlagStride = 16;   // ready flags are 16 words (one 64-byte line) apart
struct PullSource {
  const float4 *h_rows;               // the pinned rows (device-visible address)
  float4 *d_copy;                     // where the kernel leaves them in device memory (k_accum reads them there).  Without h_rows (a
                                      // device cloud): a second copy of the rows the kernel marks, for a caller that wants to keep them
  const unsigned *ready;              // [chunks * kPullFlagStride], device-visible: ready[c * kPullFlagStride] == seq once chunk c's rows are packed
  unsigned seq, chunk_shift;          // a chunk is 1 << chunk_shift rows
  unsigned long long patience_ticks;  // 100 MHz ticks a tile waits for its chunk before it flags the build (spec_miss = 2)
};

constexpr size_t kViStampWgs = 8192;   // workgroups per kernel the stamp buffer holds
struct VoxelIndex {
  DevBuf<uint32_t> bits, prefix;
  // large clouds (>= 512 k points) mark their cells by plain BYTE stores (occ[cell] = 1: idempotent, no read-modify-write, no atomics —
  // at 2 M points the bitmap's atomic ORs, one per distinct word and tile, all hitting the same few thousand words at the memory
  // side, held k_mark at 20-29 us for a 32 MB read), and a small kernel packs the bytes of the cell box into the bitmap's words
  // and clears them again (k_occ_to_bits).  W_cap * 32 bytes, zero between builds; allocated on the first large build.
  DevBuf<uint8_t> occ;
  size_t occ_words = 0;   // the bitmap capacity occ[] was allocated and cleared for
  DevBuf<unsigned long long> scan_state;  // decoupled look-back states, one per 2048-word tile
  DevBuf<unsigned> tickets;               // [1] scan tile ticket (self-resetting); [0] spare
  unsigned scan_epoch = 0;
  DevBuf<int32_t> cell_of_rank, cnt, cell_idx;
  // Large clouds (>= 512 k points) of an owner whose finalize kernel clears the rows it reads BY KEY may be summed into rows indexed
  // by the cell (k_accum<.., BY_CELL>): by_cell says how the LAST build keyed acc[] / cnt[] — row cell_of_rank[r] instead of row r.
  // Such a row is ONE 64-byte line — the ns = 4 sums and the count as a fifth 64-bit word, padded to kByCellStride words — so that a
  // tile's adds for a cell leave as one atomic instruction touching one line: one request to the memory side per (tile, cell) pair,
  // where a 32-byte row + a count array of its own sent two, and the request rate is what the flush of a 0.3 m build runs at (the
  // rank-keyed rows of VoxelFilter have the same one-line layout, vi_row_stride).  W_cap * 32 cells x 64 bytes (268 MB for the reserved 2^17-word bitmap), only while that stays below
  // kByCellMaxBytes; all zero between builds like the rank-keyed rows.  VoxelFilter (ns = 4) only.
  bool allow_by_cell = false, by_cell = false;
  bool want_cell_idx = false;   // the owner reads cell_idx[] (the linear cell of every point, -1 = none) behind the build: nn_grid.hip
  DevBuf<long long> acc;
  DevBuf<float> scratch;  // bbox partials
  // diagnostics (lsa_voxel_filter_stamps): thread 0 of every workgroup of k_mark ([0 ..)), k_accum ([kViStampWgs * 8 ..)) and k_scan ([2 kViStampWgs * 8 ..)) notes the
  // 100 MHz wall clock at its phase boundaries, 8 values per workgroup; null = off (the kernels test the pointer, nothing else)
  DevBuf<unsigned long long> stamps;
  DevBuf<DevParams> dparams;
  PinnedBuf<DevParams> h_params;
  hipEvent_t ev_done = nullptr;
  size_t W_cap = 0;       // bitmap capacity in words
  size_t acc_rows = 0;    // rows of acc / cnt known to be zero ("clean") on entry
  int acc_ns = 0;
  bool clean = false;     // bits[0..W_cap), acc, cnt are all zero
  // the owner sizes buffers of its own by the definition / the occupied-cell count: k_scan mirrors them to h_params and vi_enqueue records
  // ev_done behind it, so vi_wait() (without vi_fetch / vi_mark_done) returns while k_accum still runs
  bool mirror_at_scan = false;
  // floor of the point count the fixed-point scales are derived from: 0 = from the cloud at hand (default).  The front end
  // sets it to the capacity of its local map so that the scales are a constant of the session, and the sums of a full
  // rebuild are quantised exactly like the persistent sums of the incremental local map (local_map.hip)
  unsigned long long scale_n_min = 0;
  // speculative reuse of the previous grid definition (VoxelFilter only, see vi_enqueue)
  bool spec_valid = false, last_allow_spec = false, last_was_spec = false, last_pull_timeout = false;
  int spec_mode = 0;
  float spec_leaf[3] = {0, 0, 0};
  unsigned long long spec_n_cap = 0;
  size_t spec_W_cap = 0;
  // host copies after vi_fetch()
  size_t W = 0, nocc = 0;
  CellMap map{};
  FixScales fs{};
  void release() {
    bits.release(); prefix.release(); occ.release(); occ_words = 0; scan_state.release(); tickets.release(); cell_of_rank.release(); cnt.release(); cell_idx.release();
    acc.release(); scratch.release(); dparams.release(); h_params.release(); stamps.release();
    if (ev_done) { (void)hipEventDestroy(ev_done); ev_done = nullptr; }
    W_cap = 0; acc_rows = 0; clean = false;
  }
};

// Enqueue the whole index build for `n` points with the given rule set (mode 0 PCL / 1 NDT) and leaf:
//   bbox -> grid definition (device) -> mark bitmap -> single-pass popcount scan (+ inverse map) -> per-cell sums
//   (ns = 4 or 10): five launches.
// No host synchronisation.  Buffers are sized for n; the bitmap uses the capacity left by earlier calls (the device flags
// VI_CAPACITY if it does not fit).
// allow_spec (VoxelFilter): consecutive lidar scans have the same extent, and PCL's output ORDER (ascending i + j*dx +
// k*dx*dy) and values do not depend on where the cell box starts or how large it is, only on floor(x * inv_leaf) — so a
// build may reuse the previous build's grid definition (defined with a margin) and skip the two bounding-box launches;
// k_mark flags any finite point that falls outside (spec_miss) and the caller then rebuilds exactly.
// pull (optional, only when vi_spec_ready() holds): k_mark reads the rows from pull->h_rows as the host delivers them and copies
// them to pull->d_copy == d_pts on the way.
int vi_enqueue(lsa_context *ctx, VoxelIndex &vi, const float4 *d_pts, size_t n, int mode, const float leaf[3], int ns, bool allow_spec = false,
               const PullSource *pull = nullptr, float4 *d_keep = nullptr);
// will the next vi_enqueue(.., allow_spec = true) over n points reuse the grid definition the device holds?
bool vi_spec_ready(const VoxelIndex &vi, size_t n, int mode, const float leaf[3]);
// Allocate now what a build over up to `n` points with `ns` sums per cell will need, with a bitmap of at least w_min words
// (vi_enqueue calls this itself; a caller that knows its sizes calls it ahead so that no build allocates).
int vi_prepare(lsa_context *ctx, VoxelIndex &vi, size_t n, int ns, size_t w_min);
// Enqueue the D2H copy of the grid definition + result count, record the completion event (call after the caller's own
// finalize kernel).  vi_wait() blocks on that event only and refreshes vi.map / vi.fs / vi.W / vi.nocc.
// Returns VI_* in *status.  On VI_CAPACITY the caller calls vi_grow() and re-enqueues.
int vi_fetch(lsa_context *ctx, VoxelIndex &vi);
// Variant for callers whose own last kernel mirrors *dparams into vi.h_params.p[0] (host-mapped pinned memory): only the
// completion event is recorded, no copy kernel runs.
int vi_mark_done(lsa_context *ctx, VoxelIndex &vi);
int vi_wait(lsa_context *ctx, VoxelIndex &vi, int *status);
int vi_grow(lsa_context *ctx, VoxelIndex &vi);
// Enqueue the clean-up (zero the bitmap words / sums / counts this build touched) so the next build needs no memset.
// Runs behind the completion event, i.e. off the caller's critical path.
int vi_cleanup(lsa_context *ctx, VoxelIndex &vi, size_t n);
int vi_prepare_large(lsa_context *ctx, VoxelIndex &vi, int ns, bool with_occ);
constexpr size_t kByCellMaxBytes = (size_t)1 << 30;
constexpr size_t kByCellStride = 8;   // 64-bit words per cell-keyed row (ns = 4: four sums, the count, three spare)
// 64-bit words per row of acc[], whatever the key, the count being the word behind the ns sums (cnt[] unused): the ns = 4 rows of
// VoxelFilter are one 64-byte line each — one atomic request per (tile, cell) pair instead of two —, the ns = 10 rows of the voxel
// table two lines (sums 0..7 | sums 8, 9 and the count) instead of an unaligned 80 bytes + cnt[]: two requests per pair instead of three
constexpr size_t kRow10Stride = 16;
constexpr size_t vi_row_stride(int ns) { return ns == 4 ? kByCellStride : (ns == 10 ? kRow10Stride : (size_t)ns + 1); }

#endif
