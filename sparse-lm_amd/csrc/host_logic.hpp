// The host-side bookkeeping of the engine that touches no device: the recycling pool's ledger, the row sets of a call's
// lanes, the walk of interleaved lanes over a shared path, the launch geometry of the X^T R pass, the look-up of a row
// set's Gram, the tile numbering of the model Gram's triangle.  Kept free of HIP types so that g++ compiles it alone:
// tests/host_logic_test.cpp runs every function under AddressSanitizer + UndefinedBehaviorSanitizer in the build
// container (tools/sanitize.sh; GPU sanitizers are not available on the pool), and engine*.hip include THIS file -- what
// is tested is what runs.  (Reference counterpart: none -- the reference keeps no device state; SURVEY section 5 asks for
// the sanitizer coverage.)
#pragma once
#include <stddef.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <utility>
#include <vector>

#if defined(__HIPCC__)
#define SLM_HD __host__ __device__
#else
#define SLM_HD
#endif

namespace slm_host {

// ---------------------------------------------------------------------------------------------
// Ledger of the pool of large device blocks (engine.hip: pool_malloc / pool_free).  A freed block waits in a list of its
// exact (device, size); the next request of that shape takes it.  At most `cap` bytes wait; over the cap the blocks that
// have waited longest are released first.  The ledger never calls an allocator: `take` / `give_back` say what to do.
// ---------------------------------------------------------------------------------------------
struct PoolLedger {
  struct Idle { int dev; size_t bytes; void* block; };
  struct Live { void* block; int dev; size_t bytes; };
  std::vector<Idle> idle;  // oldest first
  std::vector<Live> live;  // pooled blocks in use
  size_t idle_bytes = 0;

  // a waiting block of exactly (dev, bytes), moved to the live list; nullptr: the caller allocates and calls adopt()
  void* take(int dev, size_t bytes) {
    for (size_t i = 0; i < idle.size(); ++i) {
      if (idle[i].dev == dev && idle[i].bytes == bytes) {
        void* p = idle[i].block;
        idle.erase(idle.begin() + (long)i);
        idle_bytes -= bytes;
        live.push_back({p, dev, bytes});
        return p;
      }
    }
    return nullptr;
  }
  void adopt(void* p, int dev, size_t bytes) { live.push_back({p, dev, bytes}); }

  // A block comes back.  Returns true when the ledger keeps it (it now waits); false: the caller releases it (not one of
  // the pool's, pooling switched off, or larger than the cap).  `evict` receives the blocks that have to make room,
  // oldest first: the caller releases them.
  bool give_back(void* p, size_t cap, bool pooling, std::vector<void*>* evict) {
    for (size_t i = 0; i < live.size(); ++i) {
      if (live[i].block != p) continue;
      const int dev = live[i].dev;
      const size_t bytes = live[i].bytes;
      live.erase(live.begin() + (long)i);
      if (!pooling || bytes > cap) return false;
      while (idle_bytes + bytes > cap && !idle.empty()) {
        evict->push_back(idle.front().block);
        idle_bytes -= idle.front().bytes;
        idle.erase(idle.begin());
      }
      idle.push_back({dev, bytes, p});
      idle_bytes += bytes;
      return true;
    }
    return false;
  }
  // the waiting blocks of a device whose last engine is going, for release -- unless a pooled block of that device is still
  // in use (a dataset outlives the engine that is being destroyed: its blocks come back later and may wait again)
  std::vector<void*> retire_device(int dev) {
    std::vector<void*> out;
    for (const Live& e : live)
      if (e.dev == dev) return out;
    for (size_t i = 0; i < idle.size();) {
      if (idle[i].dev == dev) {
        out.push_back(idle[i].block);
        idle_bytes -= idle[i].bytes;
        idle.erase(idle.begin() + (long)i);
      } else {
        ++i;
      }
    }
    return out;
  }
  // every waiting block, for release (the driver ran out of memory); the list is empty afterwards
  std::vector<void*> flush() {
    std::vector<void*> out;
    for (const Idle& e : idle) out.push_back(e.block);
    idle.clear();
    idle_bytes = 0;
    return out;
  }
};

// ---------------------------------------------------------------------------------------------
// Row sets of a call: lanes that bring the same row weights (same host pointer: the grid rows of one CV fold) and the same
// 1/n scaling share one Gram (working set, model Gram, covariance pass).  set_of[l]: the set of lane l; set_lane[s]: the
// first lane of set s.  Returns the number of sets.
// ---------------------------------------------------------------------------------------------
inline int row_sets(int n_lanes, const void* const* row_weight, const int64_t* n_eff, int* set_of, int* set_lane) {
  int n_sets = 0;
  for (int l = 0; l < n_lanes; ++l) {
    int found = -1;
    for (int m = 0; m < l && found < 0; ++m)
      if (row_weight[m] == row_weight[l] && n_eff[m] == n_eff[l]) found = set_of[m];
    if (found < 0) {
      found = n_sets;
      set_lane[n_sets++] = l;
    }
    set_of[l] = found;
  }
  return n_sets;
}

// ---------------------------------------------------------------------------------------------
// Interleaved lanes of a shared path (solve_core): lane l takes points l, l + B, l + 2 B, ... of `total` points.  The
// points beyond the last full band go to the LAST lanes (`tail_pt`), which have just solved their neighbours -- unless
// there are fewer than two full bands, where every lane simply strides to the end.
// ---------------------------------------------------------------------------------------------
struct LaneWalk {
  int32_t first;     // first point
  int32_t n_points;  // end (exclusive) of the regular walk
  int32_t stride;
  int32_t tail_pt;   // one more point after the walk, or -1
};
inline LaneWalk interleaved_walk(int lane, int B, int64_t total, bool tail_band, bool slack_deep = false) {
  LaneWalk w;
  w.first = lane;
  w.n_points = (int32_t)total;
  w.stride = B;
  w.tail_pt = -1;
  const int64_t rem = total % B, n_reg = total - rem;
  // Three passes' worth of points with at least two lanes' worth of slack (50 points on 18 lanes: 54 slots): the slack goes
  // to the lanes that hold the DEEPEST points of the first band -- the points the opening's row sample ranks least
  // reliably -- which then own that one point and nothing else: a first verification that misses there is repeated beside
  // the second band and delays nobody (as the last lanes of a uniform walk they owned the deepest point of EVERY band,
  // and a miss at the first cost the whole path a pass: eight draws of the headline's law).  The other F lanes own three
  // points each: l, l + B, and as their tail point B + F + l.
  if (slack_deep && tail_band && total > 2 * (int64_t)B && total <= 3 * (int64_t)B) {
    const int64_t e = (3 * (int64_t)B - total) / 2, F = B - e;
    if (e >= 1 && F >= 1) {
      if (lane >= F) {
        w.n_points = (int32_t)B;  // (its walk ends with its first point)
      } else {
        w.n_points = (int32_t)(B + F);
        const int64_t t = B + F + lane;
        if (t < total) w.tail_pt = (int32_t)t;
      }
      return w;
    }
  }
  if (tail_band && rem > 0 && n_reg >= 2 * (int64_t)B) {
    w.n_points = (int32_t)n_reg;
    if (lane >= B - rem) w.tail_pt = (int32_t)(n_reg + (lane - (B - rem)));
  }
  return w;
}
// points lane `lane` solves under that walk (what `expected` passes are counted from)
inline int64_t interleaved_points(const LaneWalk& w) {
  const int64_t regular = w.n_points > w.first ? ((int64_t)w.n_points - w.first + w.stride - 1) / w.stride : 0;
  return regular + (w.tail_pt >= 0 ? 1 : 0);
}

// ---------------------------------------------------------------------------------------------
// Grid of the X^T R pass on the matrix cores (launch_xtr / launch_cov_gz): column blocks of `col_block`, row blocks of a
// multiple of 8 rows, about `want` of them.
// ---------------------------------------------------------------------------------------------
struct XtrGrid { int xb, yb, rows; };
// ---------------------------------------------------------------------------------------------
// Lanes of a shared path when the caller leaves the choice to the engine (slm_solve_path_lanes, n_lanes = 0).  A
// working-set path over a large X verifies one point per lane and pass, so its passes over X are ceil(points / lanes),
// and a pass costs: sixteen lanes -- the width of the matrix cores' operand -- 1.00; seventeen to twenty the same read
// of X with the extra lanes on the vector units beside them (xtr18 / xtr20_mfma_kernel): 1.03; up to thirty-two both
// halves on the matrix cores (xtr32_mfma_kernel): 1.22; and the chain of launches between two passes about 0.45 of a
// pass whatever the count (0.58 / 0.60 / 0.71 ms and 0.25-0.3 ms at 100k x 5k).  The count with the cheapest path wins, the
// fewest lanes among equals, the points spread evenly over the passes it needs.
// `wide_ok`: lanes beyond twenty are an option -- paths in contiguous ranges (group penalties: config 3's 50 points take
// two passes on twenty-five lanes, 4.24 -> 2.79 ms).  INTERLEAVED lanes (per-feature penalties, solve_core) stop at twenty:
// beyond, a lane looks further down the path than the first working set can know -- measured on the headline shape,
// 25-32 lanes: 4 passes with 3 misses, 3.5-4.7 ms against 2.9 on eighteen (HISTORY round 5).
// `cap`: what the dataset's kernels serve (slm_dataset_max_lanes); `big`: a working-set solve over a large X at all --
// elsewhere sixteen (or what the fused kernels' table has).
// ---------------------------------------------------------------------------------------------
inline int auto_path_lanes(int64_t n_points, int cap, bool big, bool wide_ok) {
  if (n_points < 1) n_points = 1;
  const int narrow = (int)std::min<int64_t>(std::min(cap, 16), n_points);
  if (!big || cap < 20 || n_points <= 16) return narrow;
  int best = narrow;
  double best_cost = 1.45 * (double)((n_points + 15) / 16);
  auto offer = [&](int64_t lanes) {
    lanes = std::min(lanes, n_points);
    if (lanes <= 16 || lanes > cap) return;
    const int64_t passes = (n_points + lanes - 1) / lanes;
    const double cost = (double)passes * ((lanes <= 20 ? 1.03 : 1.22) + 0.45);
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = (int)lanes;
    }
  };
  // (eighteen and twenty, not the fewest lanes that make the pass count: 50 points on seventeen interleaved lanes leave a
  //  last band of sixteen that starts a whole stride up the path -- measured: a miss and a fourth pass)
  offer(18);
  offer(20);
  if (wide_ok)  // the fewest lanes beyond twenty that do it in `passes`: the points spread evenly
    for (int64_t passes = 1; passes <= (n_points + 20) / 21; ++passes) {
      const int64_t lanes = (n_points + passes - 1) / passes;
      if (lanes > 20 && lanes <= 32) offer(lanes);
    }
  return best;
}

inline int xtr_row_blocks_most(int cus, int64_t ld, int col_block) {
  const int xb = (int)((ld + col_block - 1) / col_block);
  return std::max(1, 2 * cus / xb);
}
inline XtrGrid xtr_grid(int64_t n, int64_t ld, int col_block, int64_t want) {
  XtrGrid g;
  g.xb = (int)((ld + col_block - 1) / col_block);
  if (want < 1) want = 1;
  int64_t rows = (n + want - 1) / want;
  rows = (rows + 7) / 8 * 8;
  g.rows = (int)rows;
  g.yb = (int)((n + rows - 1) / rows);  // <= want
  return g;
}

// ---------------------------------------------------------------------------------------------
// A row set's Gram among those a dataset keeps: by the two fingerprint sums of its row weights and its scaling.
// ---------------------------------------------------------------------------------------------
template <typename Entry>
inline int find_by_fingerprint(const std::vector<Entry>& entries, double fp1, double fp2, double n_eff) {
  for (size_t i = 0; i < entries.size(); ++i)
    if (entries[i].fp1 == fp1 && entries[i].fp2 == fp2 && entries[i].n_eff == n_eff) return (int)i;
  return -1;
}

// ---------------------------------------------------------------------------------------------
// Tiles of a lower triangle, row by row: tile t = (I, J <= I), t = I (I + 1) / 2 + J.  (The kernels of the model Gram
// compute the same pair from a float square root and correct it by one; this is the integer statement they are tested
// against.)
// ---------------------------------------------------------------------------------------------
inline void triangle_tile(int t, int* I_out, int* J_out) {
  int I = 0;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  *I_out = I;
  *J_out = t - I * (I + 1) / 2;
}
inline int triangle_tiles(int side) { return side * (side + 1) / 2; }
// the form the kernels use (mg_syrk_f16_*_kernel, mg_reduce_kernel, the model solver's factor assembly): a float square
// root, corrected by at most one step either way -- exact for every t the engine can ask for (tested up to 2^24)
SLM_HD inline void triangle_tile_fast(int t, int* I_out, int* J_out) {
  int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while (I * (I + 1) / 2 > t) --I;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  *I_out = I;
  *J_out = t - I * (I + 1) / 2;
}

// model Grams a dataset may keep: kept within `budget_bytes` (4 ld^2 each: fp32), at least one, at most `most`
inline int model_gram_cap(int64_t ld, double budget_bytes, int most) {
  const double each = 4.0 * (double)ld * (double)ld;  // (fp32)
  return (int)std::max<double>(1.0, std::min<double>((double)most, budget_bytes / each));
}

// ---------------------------------------------------------------------------------------------
// Every SLM_* environment knob of the engine, read ONCE (the first engine of the process; slm_reload_knobs reads again --
// tests and A/B tools that change a variable mid-process call it) into this struct: the solve loop decides on plain
// fields, not on getenv.  `get(name)` returns the variable's text or nullptr (the engine passes getenv; the sanitizer test
// a table).  None of them is needed in normal use; DESIGN.md section 7a says what each does.
// ---------------------------------------------------------------------------------------------
struct Knobs {
  // kernels and launch geometry
  bool split = true;             // SLM_SPLIT=0: no split pass
  double xtr_wgs_per_cu = 1.0;   // SLM_XTR_WGS_PER_CU (0 < f <= 2)
  bool xtr_extras = true;        // SLM_XTR_EXTRAS=0: 17-20 lanes on both matrix-core halves
  int grad_ring = -1;            // SLM_GRAD_RING: 0 off, 1 for every lane count, -1 from three lanes
  int grad_cfg[3] = {0, 0, 0};   // SLM_GRAD_CONFIG=W,C,R
  int grad_blocks_per_cu = 0;    // SLM_GRAD_BLOCKS_PER_CU (0: the table's)
  int rowdot_ring = -1;          // SLM_ROWDOT_RING: 0 / 1 forces, -1 by lane count
  bool rowdot32 = true;          // SLM_ROWDOT32=0: a read of the copy per half
  bool resid_vec = false;        // SLM_RESID_VEC=1: a row per thread
  bool resid32 = true;           // SLM_NO_RESID32: a read of the gathered columns per half
  bool cov_all_rows = false;     // SLM_COV_ALL_ROWS
  int cov_tile = 0;              // SLM_COV_TILE=3/4 (0: the packed kernel)
  int mg_syrk = -1;              // SLM_MG_SYRK=0: the register-staged product
  // step-size seeds
  int l_sketch_iters = 1;        // SLM_L_SKETCH_ITERS (1..16)
  int l_sketch_div = 32;         // SLM_L_SKETCH_DIV (1..1024)
  int power_iters = 0;           // SLM_POWER_ITERS (>= 2; 0: the caller's)
  bool l_sketch = true;          // SLM_NO_L_SKETCH
  bool sketch_cache = true;      // SLM_NO_SKETCH_CACHE
  // routes of a solve
  int ws = -1;                   // SLM_WS: 0 never, 1 from the first pass, -1 by size
  bool on_chip = true;           // SLM_ON_CHIP=0
  bool on_chip_fallback = true;  // SLM_ON_CHIP_NO_FALLBACK
  bool small_stage = true;       // SLM_NO_SMALL_STAGE
  bool wide_lanes = true;        // SLM_NO_WIDE_LANES
  int auto_lanes = 0;            // SLM_AUTO_LANES (0: the engine's choice)
  bool interleave = true;        // SLM_NO_INTERLEAVE
  bool tail_band = true;         // SLM_NO_TAIL_BAND
  bool slack_deep = true;        // SLM_NO_SLACK_DEEP
  bool carry = true;             // SLM_NO_CARRY
  bool ws_carry = true;          // SLM_NO_WS_CARRY
  bool eval_fused = false;       // SLM_EVAL_FUSED
  bool profile_unit = false;     // SLM_PROFILE_UNIT (slm_solve_opts.flags & SLM_FLAG_PROFILE_UNIT asks for the same per call)
  // working set
  bool direct = true;            // SLM_NO_DIRECT
  double ws_theta = 0.85;        // SLM_WS_THETA (0 < v <= 1)
  int ws_lookahead = 2;          // SLM_WS_LOOKAHEAD (0..64)
  int ws_append = 48;            // SLM_WS_APPEND (1..512)
  int ws_kinit = 0;              // SLM_WS_KINIT (16..512; 0: by penalty and path)
  int ws_bb = 1;                 // SLM_WS_BB
  int ws_one_solver = 0;         // SLM_WS_ONE_SOLVER
  bool hard_callwide = false;    // SLM_HARD_CALLWIDE
  int ws_power_iters = 10;       // SLM_WS_POWER_ITERS (1..40)
  int ws_miss_factor = 4;        // SLM_WS_MISS_FACTOR (1..8)
  int ws_miss_div = 8;           // SLM_WS_MISS_DIV (1..4096): coordinates outside W per further append_max after a miss
  double ws_fill = 0.0;          // SLM_WS_FILL (0.1..1; 0: by penalty and path)
  // sample start
  bool sample_start = true;      // SLM_NO_SAMPLE_START
  bool sample_start_all = false; // SLM_SAMPLE_START_ALL
  int64_t sample_min_rows = 65536;  // SLM_SAMPLE_START_MIN_ROWS (>= 64)
  int sample_div = 4;            // SLM_SAMPLE_DIV (1..64)
  // model Gram
  int mg = -1;                   // SLM_MG: 0 off, 2 forced from the first snapshot (tests), -1 by capacity
  bool mg_keep = true;           // SLM_NO_MG_KEEP
  bool handover = true;          // SLM_NO_HANDOVER
  bool light_pass = true;        // SLM_NO_LIGHT_PASS: every re-verification is a pass over X (light_kernels.hpp)
  bool lag_handover = true;      // SLM_NO_LAG_HANDOVER: a lane that has fallen behind keeps its tail point (tail_kernels.hpp)
  bool gram_owner = true;        // SLM_NO_GRAM_OWNER: every row set multiplies every row block of the gathered columns itself
  // memory, diagnostics
  double device_pool_gb = -1.0;  // SLM_DEVICE_POOL_GB (< 0: the default cap)
  bool device_pool = true;       // SLM_NO_DEVICE_POOL
  bool allow_any_arch = false;   // SLM_ALLOW_ANY_ARCH
  int trace = 0;                 // SLM_TRACE=1/2/3 (any other text: 1)
  bool trace_poll = false;       // SLM_TRACE_POLL

  template <typename Get>
  static Knobs from(Get get) {
    Knobs k;
    auto text = [&](const char* name) -> const char* { return get(name); };
    auto is_set = [&](const char* name) { return text(name) != nullptr; };
    auto first = [&](const char* name) -> char { const char* e = text(name); return e ? e[0] : '\0'; };
    auto as_int = [&](const char* name, int lo, int hi, int* out) {
      if (const char* e = text(name)) *out = std::max(lo, std::min(hi, atoi(e)));
    };
    k.split = first("SLM_SPLIT") != '0';
    if (const char* e = text("SLM_XTR_WGS_PER_CU")) {
      const double f = atof(e);
      if (f > 0.0 && f <= 2.0) k.xtr_wgs_per_cu = f;
    }
    k.xtr_extras = first("SLM_XTR_EXTRAS") != '0';
    if (const char c = first("SLM_GRAD_RING")) k.grad_ring = c == '0' ? 0 : (c == '1' ? 1 : -1);
    if (const char* e = text("SLM_GRAD_CONFIG")) {
      int W = 0, C = 0, R = 0;
      if (sscanf(e, "%d,%d,%d", &W, &C, &R) == 3) { k.grad_cfg[0] = W; k.grad_cfg[1] = C; k.grad_cfg[2] = R; }
    }
    if (const char* e = text("SLM_GRAD_BLOCKS_PER_CU")) k.grad_blocks_per_cu = std::max(1, atoi(e));
    if (const char c = first("SLM_ROWDOT_RING")) k.rowdot_ring = c == '1' ? 1 : 0;
    k.rowdot32 = first("SLM_ROWDOT32") != '0';
    k.resid_vec = first("SLM_RESID_VEC") == '1';
    k.resid32 = !is_set("SLM_NO_RESID32");
    k.cov_all_rows = is_set("SLM_COV_ALL_ROWS");
    if (const char* e = text("SLM_COV_TILE")) k.cov_tile = atoi(e);
    if (first("SLM_MG_SYRK") == '0') k.mg_syrk = 0;
    as_int("SLM_L_SKETCH_ITERS", 1, 16, &k.l_sketch_iters);
    as_int("SLM_L_SKETCH_DIV", 1, 1024, &k.l_sketch_div);
    if (const char* e = text("SLM_POWER_ITERS")) k.power_iters = std::max(2, atoi(e));
    k.l_sketch = !is_set("SLM_NO_L_SKETCH");
    k.sketch_cache = !is_set("SLM_NO_SKETCH_CACHE");
    if (const char c = first("SLM_WS")) k.ws = c == '0' ? 0 : (c == '1' ? 1 : -1);
    k.on_chip = first("SLM_ON_CHIP") != '0';
    k.on_chip_fallback = !is_set("SLM_ON_CHIP_NO_FALLBACK");
    k.small_stage = !is_set("SLM_NO_SMALL_STAGE");
    k.wide_lanes = !is_set("SLM_NO_WIDE_LANES");
    if (const char* e = text("SLM_AUTO_LANES")) k.auto_lanes = std::max(1, atoi(e));
    k.interleave = !is_set("SLM_NO_INTERLEAVE");
    k.tail_band = !is_set("SLM_NO_TAIL_BAND");
    k.slack_deep = !is_set("SLM_NO_SLACK_DEEP");
    k.carry = !is_set("SLM_NO_CARRY");
    k.ws_carry = !is_set("SLM_NO_WS_CARRY");
    k.eval_fused = is_set("SLM_EVAL_FUSED");
    k.profile_unit = is_set("SLM_PROFILE_UNIT");
    k.direct = !is_set("SLM_NO_DIRECT");
    if (const char* e = text("SLM_WS_THETA")) {
      const double v = atof(e);
      if (v > 0.0 && v <= 1.0) k.ws_theta = v;
    }
    as_int("SLM_WS_LOOKAHEAD", 0, 64, &k.ws_lookahead);
    as_int("SLM_WS_APPEND", 1, 512, &k.ws_append);
    as_int("SLM_WS_KINIT", 16, 512, &k.ws_kinit);
    if (const char* e = text("SLM_WS_BB")) k.ws_bb = atoi(e) != 0;
    if (const char* e = text("SLM_WS_ONE_SOLVER")) k.ws_one_solver = atoi(e) != 0;
    k.hard_callwide = is_set("SLM_HARD_CALLWIDE");
    as_int("SLM_WS_POWER_ITERS", 1, 40, &k.ws_power_iters);
    as_int("SLM_WS_MISS_FACTOR", 1, 8, &k.ws_miss_factor);
    as_int("SLM_WS_MISS_DIV", 1, 4096, &k.ws_miss_div);
    if (const char* e = text("SLM_WS_FILL")) k.ws_fill = std::max(0.1, std::min(1.0, atof(e)));
    k.sample_start = !is_set("SLM_NO_SAMPLE_START");
    k.sample_start_all = is_set("SLM_SAMPLE_START_ALL");
    if (const char* e = text("SLM_SAMPLE_START_MIN_ROWS")) k.sample_min_rows = std::max<int64_t>(64, atoll(e));
    as_int("SLM_SAMPLE_DIV", 1, 64, &k.sample_div);
    if (const char c = first("SLM_MG")) k.mg = c == '0' ? 0 : (c == '2' ? 2 : -1);
    k.mg_keep = !is_set("SLM_NO_MG_KEEP");
    k.handover = !is_set("SLM_NO_HANDOVER");
    k.light_pass = !is_set("SLM_NO_LIGHT_PASS");
    k.gram_owner = !is_set("SLM_NO_GRAM_OWNER");
    k.lag_handover = !is_set("SLM_NO_LAG_HANDOVER");
    if (const char* e = text("SLM_DEVICE_POOL_GB")) k.device_pool_gb = std::max(0.0, atof(e));
    k.device_pool = !is_set("SLM_NO_DEVICE_POOL");
    k.allow_any_arch = is_set("SLM_ALLOW_ANY_ARCH");
    if (const char c = first("SLM_TRACE")) k.trace = (c >= '1' && c <= '3') ? c - '0' : 1;
    k.trace_poll = is_set("SLM_TRACE_POLL");
    return k;
  }
};

}  // namespace slm_host
