// CPU test of csrc/host_logic.hpp -- the engine's device-free bookkeeping -- meant to run under AddressSanitizer and
// UndefinedBehaviorSanitizer (tools/sanitize.sh; tests/test_host_logic_cpu.py builds and runs it plainly in the CPU suite).
// Every check is against an independent statement of what the function must do; a randomised pool run models the
// allocator with malloc/free so that a double release or a leak is the sanitizers' to find.
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <random>
#include <set>
#include <string>

#include "../sparse-lm_amd/csrc/host_logic.hpp"

using namespace slm_host;

static int failures = 0;
#define CHECK(cond)                                                           \
  do {                                                                        \
    if (!(cond)) {                                                            \
      fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
      ++failures;                                                             \
    }                                                                         \
  } while (0)

static void test_pool() {
  PoolLedger book;
  std::mt19937_64 rng(1);
  const size_t sizes[] = {64u << 20, 200u << 20, 1000u << 20, 4000ull << 20};
  const size_t cap = 6000ull << 20;
  std::vector<std::pair<void*, std::pair<int, size_t>>> mine;  // blocks the "engine" holds
  std::set<void*> released;
  size_t driver_allocs = 0, reused = 0;
  for (int step = 0; step < 20000; ++step) {
    const bool alloc = mine.empty() || (rng() % 100 < 55 && mine.size() < 40);
    if (alloc) {
      const int dev = (int)(rng() % 2);
      const size_t bytes = sizes[rng() % 4];
      void* p = book.take(dev, bytes);
      if (!p) {
        p = malloc(16);  // (stands for hipMalloc of `bytes`)
        ++driver_allocs;
        book.adopt(p, dev, bytes);
      } else {
        ++reused;
      }
      for (auto& m : mine) CHECK(m.first != p);  // never handed out twice
      mine.push_back({p, {dev, bytes}});
    } else {
      const size_t i = rng() % mine.size();
      void* p = mine[i].first;
      mine.erase(mine.begin() + (long)i);
      std::vector<void*> evict;
      const bool kept = book.give_back(p, cap, true, &evict);
      for (void* q : evict) {
        CHECK(q != p);
        free(q);
      }
      if (!kept) free(p);
      CHECK(book.idle_bytes <= cap);
      size_t sum = 0;
      for (auto& e : book.idle) sum += e.bytes;
      CHECK(sum == book.idle_bytes);
    }
    CHECK(book.live.size() == mine.size());
  }
  CHECK(reused > 1000 && driver_allocs > 10);
  // a block that is not the pool's, pooling off, a block over the cap
  {
    std::vector<void*> evict;
    int dummy;
    CHECK(!book.give_back(&dummy, cap, true, &evict) && evict.empty());
    void* p = malloc(16);
    book.adopt(p, 0, 64u << 20);
    CHECK(!book.give_back(p, cap, false, &evict));
    free(p);
    p = malloc(16);
    book.adopt(p, 0, cap + 1);
    CHECK(!book.give_back(p, cap, true, &evict));
    free(p);
  }
  // retire a device: nothing while one of its blocks is in use, everything once none is
  for (auto& m : mine) {
    std::vector<void*> evict;
    if (!book.give_back(m.first, cap, true, &evict)) free(m.first);
    for (void* q : evict) free(q);
  }
  mine.clear();
  void* busy = malloc(16);
  book.adopt(busy, 1, 64u << 20);
  CHECK(book.retire_device(1).empty());
  {
    std::vector<void*> evict;
    if (!book.give_back(busy, cap, true, &evict)) free(busy);
    for (void* q : evict) free(q);
  }
  for (void* q : book.retire_device(1)) free(q);
  for (auto& e : book.idle) CHECK(e.dev == 0);
  for (void* q : book.flush()) free(q);
  CHECK(book.idle.empty() && book.idle_bytes == 0 && book.live.empty());
}

static void test_row_sets() {
  double w0[4], w1[4];
  const void* rw[16] = {w0, w0, nullptr, w1, w0, nullptr, w1, w1, w0, nullptr, nullptr, w1, w0, w0, w1, nullptr};
  int64_t ne[16];
  for (int l = 0; l < 16; ++l) ne[l] = rw[l] == w0 ? 80 : (rw[l] == w1 ? 80 : 0);
  ne[13] = 70;  // same weights, another scaling: a set of its own
  int set_of[16], set_lane[16];
  const int n = row_sets(16, rw, ne, set_of, set_lane);
  CHECK(n == 4);
  for (int l = 0; l < 16; ++l) {
    CHECK(set_of[l] >= 0 && set_of[l] < n);
    const int rep = set_lane[set_of[l]];
    CHECK(rep <= l && rw[rep] == rw[l] && ne[rep] == ne[l]);
    for (int m = 0; m < 16; ++m) CHECK((set_of[m] == set_of[l]) == (rw[m] == rw[l] && ne[m] == ne[l]));
  }
  for (int s = 0; s + 1 < n; ++s) CHECK(set_lane[s] < set_lane[s + 1]);  // in order of their first lane
  CHECK(row_sets(1, rw, ne, set_of, set_lane) == 1 && set_of[0] == 0 && set_lane[0] == 0);
}

static void test_interleaved() {
  for (int B = 1; B <= 32; ++B)  // (a shared path runs on up to thirty-two lanes)
    for (int64_t total = B; total <= 200; ++total)
      for (int tail = 0; tail < 3; ++tail) {  // (2: the slack of a three-pass path to the deepest lanes of the first band)
        std::vector<int> seen((size_t)total, 0);
        int64_t most = 0;
        for (int l = 0; l < B; ++l) {
          const LaneWalk w = interleaved_walk(l, B, total, tail != 0, tail == 2);
          int64_t mine = 0;
          for (int k = w.first; k < w.n_points; k += w.stride) {
            seen[(size_t)k] += 1;
            ++mine;
          }
          if (w.tail_pt >= 0) {
            CHECK(w.tail_pt >= w.n_points && w.tail_pt < total);
            seen[(size_t)w.tail_pt] += 1;
            ++mine;
          }
          CHECK(mine == interleaved_points(w));
          most = std::max(most, mine);
        }
        for (int64_t k = 0; k < total; ++k) CHECK(seen[(size_t)k] == 1);  // every point exactly once
        CHECK(most == (total + B - 1) / B);                               // and no lane walks more than its share
      }
  {  // 50 points on eighteen lanes: lanes 16 and 17 own points 16 and 17 alone, lane l < 16 owns l, 18 + l and 34 + l
    const LaneWalk a = interleaved_walk(16, 18, 50, true, true), b = interleaved_walk(3, 18, 50, true, true);
    CHECK(a.first == 16 && a.n_points == 18 && a.tail_pt == -1 && interleaved_points(a) == 1);
    CHECK(b.first == 3 && b.n_points == 34 && b.stride == 18 && b.tail_pt == 37 && interleaved_points(b) == 3);
  }
  const LaneWalk w = interleaved_walk(15, 16, 50, true);  // the headline: 48 regular points, 48 and 49 go to lanes 14, 15
  CHECK(w.n_points == 48 && w.tail_pt == 49 && interleaved_walk(14, 16, 50, true).tail_pt == 48 && interleaved_walk(13, 16, 50, true).tail_pt == -1);
}

static void test_auto_lanes() {
  // (points, cap, working-set solve over a large X, lanes beyond twenty allowed)
  // interleaved lanes: sixteen unless eighteen or twenty save a pass; never more than the kernels serve
  CHECK(auto_path_lanes(50, 32, true, false) == 18 && auto_path_lanes(50, 32, false, false) == 16 && auto_path_lanes(50, 16, true, false) == 16);
  CHECK(auto_path_lanes(100, 32, true, false) == 20 && auto_path_lanes(32, 32, true, false) == 16 && auto_path_lanes(36, 32, true, false) == 18);
  CHECK(auto_path_lanes(40, 32, true, false) == 20 && auto_path_lanes(41, 32, true, false) == 16 && auto_path_lanes(7, 32, true, false) == 7);
  CHECK(auto_path_lanes(1, 4, false, false) == 1 && auto_path_lanes(9, 4, false, true) == 4 && auto_path_lanes(0, 32, true, true) == 1);
  // contiguous ranges: up to thirty-two -- 50 points in two passes of twenty-five
  CHECK(auto_path_lanes(50, 32, true, true) == 25 && auto_path_lanes(60, 32, true, true) == 30 && auto_path_lanes(36, 32, true, true) == 18);
  CHECK(auto_path_lanes(64, 32, true, true) == 32 && auto_path_lanes(100, 32, true, true) == 25 && auto_path_lanes(17, 32, true, true) == 17);
  for (int64_t k = 1; k <= 400; ++k)
    for (int cap : {1, 4, 6, 16, 32})
      for (int big = 0; big < 2; ++big)
        for (int wide = 0; wide < 2; ++wide) {
          const int b = auto_path_lanes(k, cap, big != 0, wide != 0);
          const int64_t narrow = std::min<int64_t>(std::min(cap, 16), k);
          CHECK(b >= 1 && b <= cap && b <= k && b <= (wide ? 32 : 20));
          CHECK((k + b - 1) / b <= (k + narrow - 1) / narrow);  // never more passes than sixteen lanes take
          if (!big) CHECK(b == narrow);
        }
}

static void test_grid() {
  for (int cus : {1, 64, 256, 304})
    for (int64_t ld : {16, 512, 528, 5008, 10240, 16384})
      for (int64_t n : {1, 7, 8, 9, 1000, 5008, 100000, 1000003}) {
        const int most = xtr_row_blocks_most(cus, ld, 512);
        CHECK(most >= 1);
        for (int64_t want : {(int64_t)0, (int64_t)1, (int64_t)(most / 2), (int64_t)most}) {
          const XtrGrid g = xtr_grid(n, ld, 512, want);
          CHECK(g.xb * 512 >= ld && (g.xb - 1) * 512 < ld);
          CHECK(g.rows % 8 == 0 && g.rows >= 8);
          CHECK((int64_t)g.yb * g.rows >= n && (int64_t)(g.yb - 1) * g.rows < n);  // the blocks cover the rows, none is empty
          CHECK(g.yb <= std::max<int64_t>(1, want));
        }
      }
}

struct Entry { double fp1, fp2, n_eff; };
static void test_find_and_tiles() {
  std::vector<Entry> e = {{1.0, 2.0, 80.0}, {1.0, 2.0, 70.0}, {3.0, 2.0, 80.0}};
  CHECK(find_by_fingerprint(e, 1.0, 2.0, 70.0) == 1 && find_by_fingerprint(e, 3.0, 2.0, 80.0) == 2);
  CHECK(find_by_fingerprint(e, 1.0, 2.5, 80.0) == -1 && find_by_fingerprint(std::vector<Entry>(), 0, 0, 0) == -1);
  int t = 0;
  for (int I = 0; I < 300; ++I)
    for (int J = 0; J <= I; ++J, ++t) {
      int a, b, c, d;
      triangle_tile(t, &a, &b);
      triangle_tile_fast(t, &c, &d);
      CHECK(a == I && b == J && c == I && d == J);
    }
  CHECK(triangle_tiles(300) == t);
  for (int tt : {(1 << 24) - 1, 1 << 24, 8256 * 16 - 1, 33550336}) {  // far beyond any triangle the engine builds
    int a, b, c, d;
    triangle_tile(tt, &a, &b);
    triangle_tile_fast(tt, &c, &d);
    CHECK(a == c && b == d && a * (a + 1) / 2 + b == tt && b <= a);
  }
  CHECK(model_gram_cap(5008, 3.0e9, 16) == 16 && model_gram_cap(10000, 3.0e9, 16) == 7 && model_gram_cap(16384, 3.0e9, 16) == 2 &&
        model_gram_cap(100000, 3.0e9, 16) == 1);
}

// explicit sixteen-lane paths of 33 to 47 points under slack_deep (advisor, round 5): 48 slots, the slack goes to the lanes
// that hold the deepest points of the first band -- they own that one point -- and every point is owned exactly once
static void test_interleaved_sixteen_lanes() {
  for (int total = 33; total <= 47; ++total) {
    std::vector<int> owner(total, -1);
    const int64_t e = (48 - total) / 2, F = 16 - e;
    for (int lane = 0; lane < 16; ++lane) {
      const LaneWalk w = interleaved_walk(lane, 16, total, true, true);
      std::vector<int> pts;
      for (int q = w.first; q < w.n_points; q += w.stride) pts.push_back(q);
      if (w.tail_pt >= 0) pts.push_back(w.tail_pt);
      CHECK((int64_t)pts.size() == interleaved_points(w));
      if (e >= 1) {
        if (lane >= F) CHECK(pts.size() == 1 && pts[0] == lane);  // the deepest points of the first band: one point each
        else CHECK(pts.size() >= 2 && pts[0] == lane && pts[1] == 16 + lane);
      }
      for (int q : pts) {
        CHECK(q >= 0 && q < total && owner[q] == -1);
        if (q >= 0 && q < total) owner[q] = lane;
      }
    }
    for (int q = 0; q < total; ++q) CHECK(owner[q] >= 0);
    if (total == 40) {  // (the advisor's example: e = 4, lanes 12..15 own a single point; lanes 0..11 own l, 16 + l, 28 + l)
      CHECK(interleaved_walk(12, 16, 40, true, true).tail_pt == -1 && interleaved_walk(12, 16, 40, true, true).n_points == 16);
      CHECK(interleaved_walk(0, 16, 40, true, true).tail_pt == 28 && interleaved_walk(11, 16, 40, true, true).tail_pt == 39);
    }
  }
}

// the SLM_* knobs: read once into a struct (round-5 verdict, item 5) -- defaults, every kind of field, clamping, reload
static void test_knobs() {
  std::map<std::string, std::string> env;
  auto get = [&](const char* name) -> const char* {
    auto it = env.find(name);
    return it == env.end() ? nullptr : it->second.c_str();
  };
  const Knobs d = Knobs::from(get);  // nothing set: what normal use runs with
  CHECK(d.split && d.xtr_extras && d.rowdot32 && d.resid32 && d.on_chip && d.wide_lanes && d.interleave && d.carry && d.ws_carry);
  CHECK(d.ws == -1 && d.mg == -1 && d.grad_ring == -1 && d.rowdot_ring == -1 && d.auto_lanes == 0 && d.trace == 0);
  CHECK(d.ws_theta == 0.85 && d.ws_lookahead == 2 && d.ws_append == 48 && d.ws_kinit == 0 && d.ws_fill == 0.0 && d.ws_power_iters == 10);
  CHECK(d.sample_start && !d.sample_start_all && d.sample_min_rows == 65536 && d.sample_div == 4 && d.l_sketch_div == 32 && d.l_sketch_iters == 1);
  CHECK(d.device_pool && d.device_pool_gb < 0.0 && !d.allow_any_arch && d.xtr_wgs_per_cu == 1.0 && d.direct && d.mg_keep && d.handover);
  CHECK(!d.profile_unit && !d.eval_fused && d.power_iters == 0 && d.grad_cfg[0] == 0);
  CHECK(d.light_pass && d.gram_owner && d.lag_handover && d.ws_miss_factor == 4 && d.ws_miss_div == 8);  // (round 6)
  env["SLM_WS"] = "0"; env["SLM_MG"] = "2"; env["SLM_TRACE"] = "3"; env["SLM_TRACE_POLL"] = "1"; env["SLM_SPLIT"] = "0";
  env["SLM_GRAD_CONFIG"] = "8,5,2"; env["SLM_WS_THETA"] = "0.7"; env["SLM_WS_APPEND"] = "9999"; env["SLM_WS_KINIT"] = "3";
  env["SLM_SAMPLE_DIV"] = "0"; env["SLM_SAMPLE_START_MIN_ROWS"] = "10"; env["SLM_XTR_WGS_PER_CU"] = "7"; env["SLM_NO_CARRY"] = "";
  env["SLM_DEVICE_POOL_GB"] = "-3"; env["SLM_ROWDOT_RING"] = "1"; env["SLM_GRAD_RING"] = "0"; env["SLM_AUTO_LANES"] = "20";
  env["SLM_WS_FILL"] = "5"; env["SLM_POWER_ITERS"] = "1"; env["SLM_ON_CHIP"] = "0";
  env["SLM_NO_LIGHT_PASS"] = "1"; env["SLM_NO_GRAM_OWNER"] = ""; env["SLM_NO_LAG_HANDOVER"] = "1"; env["SLM_WS_MISS_DIV"] = "0"; env["SLM_WS_MISS_FACTOR"] = "99";
  Knobs k = Knobs::from(get);
  CHECK(!k.light_pass && !k.gram_owner && !k.lag_handover && k.ws_miss_div == 1 && k.ws_miss_factor == 8);
  CHECK(k.ws == 0 && k.mg == 2 && k.trace == 3 && k.trace_poll && !k.split && !k.on_chip);
  CHECK(k.grad_cfg[0] == 8 && k.grad_cfg[1] == 5 && k.grad_cfg[2] == 2 && k.ws_theta == 0.7);
  CHECK(k.ws_append == 512 && k.ws_kinit == 16 && k.sample_div == 1 && k.sample_min_rows == 64);  // clamped to their ranges
  CHECK(k.xtr_wgs_per_cu == 1.0);  // (outside (0, 2]: ignored)
  CHECK(!k.carry && k.ws_carry && k.device_pool_gb == 0.0 && k.rowdot_ring == 1 && k.grad_ring == 0 && k.auto_lanes == 20);
  CHECK(k.ws_fill == 1.0 && k.power_iters == 2);
  env["SLM_WS"] = "1"; env["SLM_MG"] = "0"; env["SLM_TRACE"] = "yes"; env["SLM_WS_THETA"] = "1.5"; env.erase("SLM_SPLIT");
  k = Knobs::from(get);  // a second read follows the environment (slm_reload_knobs)
  CHECK(k.ws == 1 && k.mg == 0 && k.trace == 1 && k.ws_theta == 0.85 && k.split);
}

int main() {
  test_pool();
  test_row_sets();
  test_interleaved();
  test_interleaved_sixteen_lanes();
  test_auto_lanes();
  test_grid();
  test_find_and_tiles();
  test_knobs();
  if (failures) {
    fprintf(stderr, "host_logic_test: %d check(s) failed\n", failures);
    return 1;
  }
  printf("host_logic_test: ok\n");
  return 0;
}
