// local_map.h — internal: the front end's INCREMENTAL sliding local map (local_map.hip).
//
// Reference being replaced: FrontEnd::UpdateWithNewFrame (lidar_localization/src/mapping/front_end/front_end.cpp:348-424):
// every key-frame it re-transforms the <= 20 key-frame clouds, concatenates them (~2.4 M points), filters the result at the
// local-map leaf (0.3 m; once >= 10 key-frames) and hands it to SetInputTarget, which rebuilds the whole NDT voxel table.
// The full-rebuild path of this library does the same on the device (front_end.hip: ~310 us per key-frame, two host round trips).
//
// Here the two aggregation levels are PERSISTENT and a key-frame touches only what changes:
//   level F (filter, leaf 0.3 m): per occupied cell the exact int64 fixed-point sums of (p - corner) and the intensity over
//            the points of all key-frames in the window, and the cell's current output point (its f32 mean);
//   level N (NDT, leaf 1.0 m):    per occupied cell the 10 fixed-point sums over the level-F output points (or, while the map
//            is still unfiltered, over the raw points), the 80-byte voxel record and the dense probe table entry.
// A new key-frame ADDS its points to level F, the key-frame it evicts SUBTRACTS its own; every F cell touched re-emits its
// output point and moves the difference (old point out, new point in) into level N; every N cell touched is re-finalised.
// Integer sums are exact and order-free, so after any sequence of adds and subtracts the sums — and therefore every output
// point, record and table entry — are bit for bit what a rebuild from the current window gives, PROVIDED both quantise alike:
// the fixed-point scales are a constant of the session (VoxelIndex::scale_n_min / vi_make_scales).
//
// Cells are keyed by ABSOLUTE integer coordinates (floor(x * inv_leaf) resp. floor(x / leaf): what PCL's VoxelGrid and the
// reference's VoxelGrid compute before subtracting their bounding-box minimum) in open-addressing hash tables; a cell whose
// count returns to zero keeps its key (it is usually next to live ones and often comes back), tables are rebuilt when keys
// (live + dead) pass 60 % of the capacity.  The NDT pass probes a dense table over a WINDOW of level-N cells that contains
// the map's bounding box with a margin; cells outside the bounding box are empty in the reference too, so the window's
// extent is not observable.  A valid voxel outside the window raises a flag that the front end reads before its next
// ScanMatch and re-fits the window.
//
// Layout (round 4).  ONE cache line per level-F cell (64 B: key, 4 sums, count word, output point) and one 128-byte line per
// level-N cell (key, 10 sums, count word, state), so that everything a kernel needs of a cell comes with one request and the
// NS + 1 atomics of a (tile, cell) contribution are ONE wave-instruction's adjacent lanes into one line.  The count word's high
// half counts the contributions of the running update: the returning add that carries a tile's count tells the tile whether it
// is the cell's first toucher — it then lists the cell as dirty — with no flag array and no exchange of its own.  Dirty cells are
// listed per TILE (a region of the list per workgroup of the producing kernel, its length in a per-region word): no counter that
// every workgroup appends to; the consuming kernel scans the region lengths (a few hundred words) and deals the entries out
// evenly.  What the three launches of a key-frame cost before (23 + 22 + 14 us, profiles/r03_*) and after: DESIGN.md 6d.
#ifndef LSA_LOCAL_MAP_H_
#define LSA_LOCAL_MAP_H_

#include "voxel_index.h"

struct LmStatus {
  unsigned keys_f, keys_n;            // keys in the tables (live + dead)
  int live_f, live_n;                 // cells with a positive count
  unsigned overflow;                  // 1: a probe sequence ran out of slots (rebuild larger); 2: coordinates beyond +-2^20 cells
  unsigned window_miss;               // a valid voxel lies outside the dense window
  unsigned done;                      // workgroups of k_lm_ndt_dirty that have finished (the last one publishes the status)
  unsigned pad;
  unsigned long long epoch;           // which update this snapshot is of
  int bmin[3], bmax[3];               // k_lm_bounds: cell bounds of the valid voxels
};

// count word of a cell: bits 0..31 the point count (never negative at any instant: points leave only after they came),
// bits 32..62 the contributions received in the running update (0 = clean), bit 63 (level F) "has an output point in level N"
constexpr unsigned long long kLmTouch = 1ull << 32;
constexpr unsigned long long kLmHasOut = 1ull << 63;

struct alignas(64) LmCellF {
  unsigned long long key;             // packed absolute cell (21 bits per axis), 0 = empty
  long long sum[4];
  unsigned long long cnt;
  float4 out;                         // the cell's current output point (valid with kLmHasOut)
};
struct alignas(128) LmCellN {
  unsigned long long key;
  long long sum[10];
  unsigned long long cnt;
  int32_t state, alive;               // record state (0 / 1 / -1); cell counted in live_n
  unsigned long long spare[3];
};
static_assert(sizeof(LmCellF) == 64 && sizeof(LmCellN) == 128, "one line per cell");

// a dirty list: `regions` regions of `region_cap` slots + one overflow region behind them; len[r] = entries of region r,
// written (also when 0) by the workgroup that owns the region in the producing launch
struct LmDirty {
  DevBuf<unsigned> slot;
  DevBuf<unsigned> len;
  unsigned regions = 0, region_cap = 0, over_cap = 0;
  void release() { slot.release(); len.release(); regions = region_cap = over_cap = 0; }
};

struct LocalMap {
  lsa_context *ctx = nullptr;
  DevBuf<LmCellF> f;
  DevBuf<LmCellN> n;
  unsigned cap_f = 0, cap_n = 0;      // powers of two
  LmDirty dirty_f, dirty_n;
  DevBuf<double> rec;                 // level N: [cap][10] voxel records, indexed by hash slot
  DevBuf<float4> dense;               // dense probe table over the window
  int wmin[3] = {0, 0, 0}, wdim[3] = {0, 0, 0};
  bool window_set = false;
  DevBuf<LmStatus> d_status;
  PinnedBuf<LmStatus> h_status;
  DevBuf<unsigned long long> stamps;  // diagnostics (lm_stamps_enable): 3 kernels x kMaxRegions workgroups x 8 clock values
  FixScales fs_f{}, fs_n{};
  float leaf_f[3] = {0.3f, 0.3f, 0.3f}, leaf_n = 1.0f;
  int pcl = 0;
  bool filtered = false;              // level F in use (>= 10 key-frames and a local-map filter configured)
  bool valid = false;                 // the persistent sums are those of the current window
  bool pending = false;               // an update has been enqueued whose status the owner has not looked at yet
  bool n_zero = false, f_zero = false; // the level's table is known to be all zero (fresh from lm_reserve): the next rebuild need not clear it
  unsigned long long epoch = 0;
  long long dense_max_cells = 16ll << 20;
  unsigned long long rebuilds = 0, refits = 0, updates = 0;
  void release() {
    f.release(); n.release(); dirty_f.release(); dirty_n.release(); rec.release(); dense.release();
    d_status.release(); h_status.release(); stamps.release();
    cap_f = cap_n = 0;
    valid = false; window_set = false;
  }
};

// scales / leaves / mode; capacities are chosen here (and grown by lm_rebuild when a table fills up)
int lm_configure(LocalMap &lm, lsa_context *ctx, const float leaf_f[3], float leaf_n, int pcl, unsigned long long scale_n);
// optional: allocate now, at the default capacities, what a session otherwise allocates on the way (local_map.hip)
int lm_reserve(LocalMap &lm, lsa_context *ctx, bool with_level_f);
// The steady-state key-frame in ONE launch: the slot's old contents (the evicted key-frame's transformed points; skipped if
// !evict) leave the sums while the new key-frame's cloud d_raw is transformed (row-major 3x4, f32, PCL's order) into the slot
// and enters them; a cell both clouds touch receives its net change once.  Enqueue only.
// d_keep (optional): the key-frame's own copy of its cloud, written on the way (static weights set to 1 if force_w_one,
// front_end.cpp:304-306).
int lm_swap(LocalMap &lm, float4 *d_slot, size_t slot_cap, bool evict, const float4 *d_raw, size_t n_raw, const float T_rowmajor12[12],
            float4 *d_keep = nullptr, bool force_w_one = false);
// After lm_swap: move the touched level-F cells' changes into level N, re-finalise the touched level-N cells, publish the
// status.  Enqueue only; the owner reads lm.h_status once later stream work has completed.
int lm_finish(LocalMap &lm);
// the same in two halves (lm_finish = begin + end): everything that writes what a registration pass reads — records, probe table —
// is in the second
int lm_finish_begin(LocalMap &lm);
int lm_finish_end(LocalMap &lm);
// Everything from scratch from `nranges` ranges of the local-map buffer (first key-frame, the switch to the filtered map,
// a table that filled up, a re-laid-out buffer): clears, re-adds, finalises every cell, fits the window.  Waits for the stream.
// leave_in_flight: the window's last two kernels (clear + fill of the dense table) are not waited for — the caller orders the table's
// readers behind lm.ctx->stream itself (an event); everything the host reads has arrived by then either way
int lm_rebuild(LocalMap &lm, const float4 *const *d_ranges, const size_t *n_ranges, int nranges, bool filtered, bool leave_in_flight = false);
// Re-fit the dense window around the valid voxels (waits for the stream).  LSA_ERR_CAPACITY: the window would exceed dense_max_cells.
int lm_refit_window(LocalMap &lm);
// Look at the status of the last update (the caller guarantees its kernels have completed): 0 fine, 1 window re-fit needed,
// 2 rebuild needed (tables filling up / overflow)
int lm_check(LocalMap &lm);

// diagnostics: the three kernels of the updates that follow note the 100 MHz wall clock at their phase boundaries (8 values per
// workgroup: local_map.hip lm_stamp); fetch copies the 3 x *per_kernel values out and clears them
int lm_stamps_enable(LocalMap &lm, bool on);
int lm_stamps_fetch(LocalMap &lm, unsigned long long *out, size_t cap, size_t *per_kernel);

// tests: compare level N with a full table build of the same window (its occupied cells, records and states).
// out[0] cells of the full table compared, [1] cells that differ in any bit, [2] live cells of level N, [3] valid voxels of level N
int lm_verify(LocalMap &lm, const int32_t *full_cell_of_rank, const double *full_rec, const int32_t *full_state, size_t V, const CellMap &map,
              unsigned long long out[4]);

#endif
