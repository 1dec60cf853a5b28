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
// ScanMatch (behind the frame filter's wait, which orders it after this update's kernels) and re-fits the window.
#ifndef LSA_LOCAL_MAP_H_
#define LSA_LOCAL_MAP_H_

#include "voxel_index.h"

struct LmStatus {
  unsigned n_dirty_f, n_dirty_n;      // dirty lists being built
  unsigned keys_f, keys_n;            // keys in the tables (live + dead)
  int live_f, live_n;                 // cells with a positive count
  unsigned overflow;                  // a probe sequence ran out of slots: the tables must be rebuilt larger
  unsigned window_miss;               // a valid voxel lies outside the dense window
  unsigned done;                      // workgroups of k_lm_ndt_dirty that have finished (the last one publishes the status)
  unsigned pad;
  unsigned long long epoch;           // which update this snapshot is of
  int bmin[3], bmax[3];               // k_lm_bounds: cell bounds of the valid voxels
};

struct LmHash {
  DevBuf<unsigned long long> keys;    // packed absolute cell (21 bits per axis); ~0 = empty
  DevBuf<int32_t> cnt;
  DevBuf<long long> acc;              // [cap][ns]
  DevBuf<unsigned> flag;              // 1 = in the dirty list
  DevBuf<unsigned> dirty;             // slots
  unsigned cap = 0;                   // power of two
  int ns = 0;
  void release() { keys.release(); cnt.release(); acc.release(); flag.release(); dirty.release(); cap = 0; }
};

struct LocalMap {
  lsa_context *ctx = nullptr;
  LmHash f, n;
  DevBuf<float4> out;                 // level F: the cell's current output point
  DevBuf<int32_t> has_out;            // level F: 1 if `out` is in level N
  DevBuf<double> rec;                 // level N: [cap][10] voxel records, indexed by hash slot
  DevBuf<int32_t> state, alive;       // level N: record state (0 / 1 / -1), cell counted in live_n
  DevBuf<float4> dense;               // dense probe table over the window
  int wmin[3] = {0, 0, 0}, wdim[3] = {0, 0, 0};
  bool window_set = false;
  DevBuf<LmStatus> d_status;
  PinnedBuf<LmStatus> h_status;
  FixScales fs_f{}, fs_n{};
  float leaf_f[3] = {0.3f, 0.3f, 0.3f}, leaf_n = 1.0f;
  int pcl = 0;
  bool filtered = false;              // level F in use (>= 10 key-frames and a local-map filter configured)
  bool valid = false;                 // the persistent sums are those of the current window
  bool pending = false;               // an update has been enqueued whose status the owner has not looked at yet
  unsigned long long epoch = 0;
  long long dense_max_cells = 16ll << 20;
  unsigned long long rebuilds = 0, refits = 0, updates = 0;
  void release() {
    f.release(); n.release(); out.release(); has_out.release(); rec.release(); state.release(); alive.release(); dense.release();
    d_status.release(); h_status.release();
    valid = false; window_set = false;
  }
};

// scales / leaves / mode; capacities are chosen here (and grown by lm_rebuild when a table fills up)
int lm_configure(LocalMap &lm, lsa_context *ctx, const float leaf_f[3], float leaf_n, int pcl, unsigned long long scale_n);
// One range of the local-map buffer (a slot of transformed points, NaN padded) added to (sign +1) or subtracted from
// (sign -1) the persistent sums.  Enqueue only.
int lm_points(LocalMap &lm, const float4 *d_pts, size_t n, int sign);
// The steady-state key-frame in ONE launch: the slot's old contents (the evicted key-frame's transformed points; skipped if
// !evict) leave the sums while the new key-frame's cloud d_raw is transformed (row-major 3x4, f32, PCL's order) into the slot
// and enters them; a cell both clouds touch receives its net change once.  Enqueue only.
// d_keep (optional): the key-frame's own copy of its cloud, written on the way (static weights set to 1 if force_w_one,
// front_end.cpp:304-306).
int lm_swap(LocalMap &lm, float4 *d_slot, size_t slot_cap, bool evict, const float4 *d_raw, size_t n_raw, const float T_rowmajor12[12],
            float4 *d_keep = nullptr, bool force_w_one = false);
// After the adds / subtracts of an update: move the touched level-F cells' changes into level N, re-finalise the touched
// level-N cells, publish the status.  Enqueue only; the owner reads lm.h_status once later stream work has completed.
int lm_finish(LocalMap &lm);
// Everything from scratch from `nranges` ranges of the local-map buffer (first key-frame, the switch to the filtered map,
// a table that filled up, a re-laid-out buffer): clears, re-adds, finalises every cell, fits the window.  Waits for the stream.
int lm_rebuild(LocalMap &lm, const float4 *const *d_ranges, const size_t *n_ranges, int nranges, bool filtered);
// Re-fit the dense window around the valid voxels (waits for the stream).  LSA_ERR_CAPACITY: the window would exceed dense_max_cells.
int lm_refit_window(LocalMap &lm);
// Look at the status of the last update (the caller guarantees its kernels have completed): 0 fine, 1 window re-fit needed,
// 2 rebuild needed (tables filling up / overflow)
int lm_check(LocalMap &lm);

// tests: compare level N with a full table build of the same window (its occupied cells, records and states).
// out[0] cells of the full table compared, [1] cells that differ in any bit, [2] live cells of level N, [3] valid voxels of level N
int lm_verify(LocalMap &lm, const int32_t *full_cell_of_rank, const double *full_rec, const int32_t *full_state, size_t V, const CellMap &map,
              unsigned long long out[4]);

#endif
