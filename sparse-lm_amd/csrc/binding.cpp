// Compiled binding of the hot calls of include/slm_engine.h (pybind11): slm_solve_lanes, slm_solve_path_lanes and
// slm_dataset_create, with the marshalling of a call -- path points and their secant factors, penalty vectors, warm starts,
// row masks, result blocks -- done here instead of in Python, and the GIL released while the engine works.  This is the
// "thin pybind11 C-ABI" layer the estimators' `_solve` seam goes through (reference src/sparselm/model/_base.py:512-519:
// the cvxpy `problem.solve(...)` call); sparselm_amd/_engine.py keeps its ctypes binding of EVERY entry point -- the ABI's
// test harness, and the route of everything that is not hot.  Plain C ABI underneath: nothing here knows the engine's types.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/slm_engine.h"

namespace py = pybind11;
using arr_d = py::array_t<double, py::array::c_style | py::array::forcecast>;

namespace {

py::object g_engine_error, g_nonfinite_error;  // sparselm_amd._engine.EngineError / NonFiniteError

// status code -> the exception the ctypes layer raises for it (_engine._check)
void check(int rc) {
  if (rc == SLM_OK) return;
  const char* m = slm_last_error();
  const std::string msg = m ? m : "";
  switch (rc) {
    case SLM_ERR_BAD_ARG: throw py::value_error(msg);
    case SLM_ERR_OOM: PyErr_SetString(PyExc_MemoryError, msg.c_str()); throw py::error_already_set();
    case SLM_ERR_UNSUPPORTED: PyErr_SetString(PyExc_NotImplementedError, msg.c_str()); throw py::error_already_set();
    case SLM_ERR_NON_FINITE:
      if (g_nonfinite_error) {
        PyErr_SetString(g_nonfinite_error.ptr(), msg.c_str());
        throw py::error_already_set();
      }
      [[fallthrough]];
    default: {
      const std::string full = "[slm status " + std::to_string(rc) + "] " + msg;
      if (g_engine_error) {
        PyErr_SetString(g_engine_error.ptr(), full.c_str());
        throw py::error_already_set();
      }
      throw std::runtime_error(full);
    }
  }
}

// secant factors gamma_k = (s_k - s_{k-1}) / (s_{k-1} - s_{k-2}) of a path whose points are multiples s_k of one penalty
// direction; zeros otherwise (the rule of _engine.path_extrapolation)
void extrapolation(const double* pts, int64_t K, double* gam) {
  for (int64_t k = 0; k < K; ++k) gam[k] = 0.0;
  if (K < 3) return;
  int64_t best = 0;
  double top = -1.0, big = 0.0;
  for (int64_t k = 0; k < K; ++k) {
    const double v = std::fabs(pts[3 * k]) + std::fabs(pts[3 * k + 1]) + std::fabs(pts[3 * k + 2]);
    if (v > top) {
      top = v;
      best = k;
    }
    for (int c = 0; c < 3; ++c) big = std::max(big, std::fabs(pts[3 * k + c]));
  }
  const double* ref = pts + 3 * best;
  const double nrm = ref[0] * ref[0] + ref[1] * ref[1] + ref[2] * ref[2];
  if (!(nrm > 0.0)) return;
  std::vector<double> s((size_t)K);
  double off = 0.0;
  for (int64_t k = 0; k < K; ++k) {
    s[(size_t)k] = (pts[3 * k] * ref[0] + pts[3 * k + 1] * ref[1] + pts[3 * k + 2] * ref[2]) / nrm;
    for (int c = 0; c < 3; ++c) off = std::max(off, std::fabs(s[(size_t)k] * ref[c] - pts[3 * k + c]));
  }
  if (off > 1e-12 * big) return;  // the penalty changes shape along the path: a prediction would be meaningless
  for (int64_t k = 2; k < K; ++k) {
    const double den = s[(size_t)k - 1] - s[(size_t)k - 2], num = s[(size_t)k] - s[(size_t)k - 1];
    if (den == 0.0) continue;
    const double g = num / den;
    if (std::isfinite(g) && std::fabs(g) <= 10.0) gam[k] = g;
  }
}

// what a call keeps alive until the engine returns: converted arrays, broadcast vectors
struct Keep {
  std::vector<py::object> objs;
  std::vector<std::vector<double>> vecs;
  // one conversion per (Python object, length): lanes that share a row mask hand the engine the SAME pointer -- one
  // check, one upload, one Gram
  std::unordered_map<PyObject*, std::pair<int64_t, const double*>> seen;
};

// None -> nullptr; an array of `size` doubles; a scalar / one-element array -> `size` copies of it
const double* vector_arg(const py::handle& v, int64_t size, const char* name, Keep& keep) {
  if (v.is_none()) return nullptr;
  auto hit = keep.seen.find(v.ptr());
  if (hit != keep.seen.end() && hit->second.first == size) return hit->second.second;
  arr_d a = arr_d::ensure(v);
  if (!a) throw py::value_error(std::string(name) + " is not convertible to float64");
  const double* out = nullptr;
  if (a.size() == size) {
    out = a.data();
    keep.objs.push_back(a);
  } else if (a.size() == 1) {
    keep.vecs.emplace_back((size_t)size, *a.data());
    out = keep.vecs.back().data();
  } else {
    throw py::value_error(std::string(name) + " has " + std::to_string(a.size()) + " entries, expected " + std::to_string(size));
  }
  keep.seen[v.ptr()] = {size, out};
  return out;
}

py::object dict_get(const py::dict& d, const char* key) {
  py::str k(key);
  return d.contains(k) ? py::reinterpret_borrow<py::object>(d[k]) : py::none();
}

py::tuple stats_tuple(const slm_solve_stats& st) {
  return py::make_tuple(st.grad_launches, st.grad_timed, st.grad_ms_total, st.wall_ms, st.lipschitz_ms, st.ws_builds, st.ws_appends,
                        st.ws_refined, st.ws_misses, st.ws_columns, st.ws_inner_iters, st.ws_direct_steps, st.mg_rounds, st.mg_inner_iters,
                        st.mg_rejected, st.mg_build_ms, st.light_passes, st.light_columns);
}

py::array info_bytes(int64_t count) {
  return py::array_t<uint8_t>((py::ssize_t)(count * (int64_t)sizeof(slm_point_info)));
}

// slm_solve_lanes: `specs` is the list of lane dicts of Dataset.solve_lanes; `alloc(n)` hands out a float64 block of n
// entries (the binding's page-locked result pool).  Returns (betas block, group-norm block or None, records as bytes,
// points per lane, stats).
// `reweighted`: every spec carries "reweight" = (coef_scale, group_scale or None, numerator, eps, tol, n_coef, n_group) and the
// call is slm_solve_lanes_reweighted; the tuple then ends with the rounds run per lane.
py::tuple solve_lanes_any(uintptr_t ds, const py::list& specs, int64_t n, int64_t p, int64_t G, double tol, int max_iter, int check_every,
                          double L, uint32_t flags, bool want_gn, bool extrapolate, const py::object& alloc, bool reweighted) {
  const int nl = (int)specs.size();
  std::vector<slm_reweight> rules(reweighted ? (size_t)nl : 0);
  std::vector<int32_t> rounds(reweighted ? (size_t)nl : 0, 0);
  if (nl < 1 || nl > SLM_MAX_CELLS) throw py::value_error("between 1 and " + std::to_string(SLM_MAX_CELLS) + " lanes, got " + std::to_string(nl));
  Keep keep;
  std::vector<slm_lane> lanes((size_t)nl);
  std::vector<slm_penalty> pens((size_t)nl);
  std::vector<std::vector<slm_path_point>> points((size_t)nl);
  std::vector<int64_t> ks((size_t)nl);
  int64_t total = 0;
  for (int l = 0; l < nl; ++l) {
    const py::dict spec = py::reinterpret_borrow<py::dict>(specs[(size_t)l]);
    arr_d pts = arr_d::ensure(dict_get(spec, "points"));
    if (!pts || pts.size() % 3 != 0 || pts.size() == 0) throw py::value_error("lane " + std::to_string(l) + ": points must be (K, 3)");
    const int64_t K = pts.size() / 3;
    std::vector<double> gam((size_t)K, 0.0);
    const py::object ex = dict_get(spec, "extrap");
    if (!ex.is_none()) {
      arr_d e = arr_d::ensure(ex);
      if (!e || e.size() != K) throw py::value_error("extrap has the wrong length");
      std::memcpy(gam.data(), e.data(), sizeof(double) * (size_t)K);
    } else if (extrapolate) {
      extrapolation(pts.data(), K, gam.data());
    }
    points[(size_t)l].resize((size_t)K);
    for (int64_t k = 0; k < K; ++k) points[(size_t)l][(size_t)k] = slm_path_point{pts.data()[3 * k], pts.data()[3 * k + 1], pts.data()[3 * k + 2], gam[(size_t)k]};
    pens[(size_t)l].a = vector_arg(dict_get(spec, "a"), p, "a", keep);
    pens[(size_t)l].b = vector_arg(dict_get(spec, "b"), G, "b", keep);
    pens[(size_t)l].d = vector_arg(dict_get(spec, "d"), G, "d", keep);
    slm_lane& ln = lanes[(size_t)l];
    std::memset(&ln, 0, sizeof(ln));
    ln.pen = &pens[(size_t)l];
    ln.points = points[(size_t)l].data();
    ln.n_points = (int32_t)K;
    ln.beta0 = vector_arg(dict_get(spec, "beta0"), p, "beta0", keep);
    ln.row_weight = vector_arg(dict_get(spec, "row_weight"), n, "row_weight", keep);
    const py::object ne = dict_get(spec, "n_eff");
    ln.n_eff = ne.is_none() ? 0 : ne.cast<int64_t>();
    ks[(size_t)l] = K;
    total += K;
    if (reweighted) {
      const py::object rw = dict_get(spec, "reweight");
      if (rw.is_none()) throw py::value_error("lane " + std::to_string(l) + ": no re-weighting rule");
      const py::tuple t = rw.cast<py::tuple>();
      if (t.size() != 7) throw py::value_error("reweight = (coef_scale, group_scale, numerator, eps, tol, n_coef, n_group)");
      slm_reweight& r = rules[(size_t)l];
      r.coef_scale = t[0].cast<double>();
      r.n_coef = t[5].cast<int32_t>();
      r.n_group = t[6].cast<int32_t>();
      r.group_scale = vector_arg(t[1], r.n_group, "group_scale", keep);
      r.numerator = t[2].cast<double>();
      r.eps = t[3].cast<double>();
      r.tol = t[4].cast<double>();
    }
  }
  py::array betas = alloc(total * p).cast<py::array>();
  py::object gn_obj = py::none();
  double* gn_ptr = nullptr;
  if (want_gn) {
    py::array g = alloc(total * G).cast<py::array>();
    gn_ptr = static_cast<double*>(g.mutable_data());
    gn_obj = g;
  }
  py::array infos = info_bytes(total);
  auto* inf = static_cast<slm_point_info*>(infos.mutable_data());
  std::memset(inf, 0, sizeof(slm_point_info) * (size_t)total);
  double* bp = static_cast<double*>(betas.mutable_data());
  int64_t at = 0;
  for (int l = 0; l < nl; ++l) {
    lanes[(size_t)l].betas_out = bp + at * p;
    lanes[(size_t)l].group_norms_out = gn_ptr ? gn_ptr + at * G : nullptr;
    lanes[(size_t)l].infos = inf + at;
    at += ks[(size_t)l];
  }
  slm_solve_opts opts{tol, max_iter, check_every, L, flags};
  slm_solve_stats st;
  std::memset(&st, 0, sizeof(st));
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = reweighted ? slm_solve_lanes_reweighted(reinterpret_cast<slm_dataset*>(ds), lanes.data(), rules.data(), nl, &opts, &st, rounds.data())
                    : slm_solve_lanes(reinterpret_cast<slm_dataset*>(ds), lanes.data(), nl, &opts, &st);
  }
  check(rc);
  py::list kl;
  for (int64_t k : ks) kl.append(k);
  if (!reweighted) return py::make_tuple(betas, gn_obj, infos, kl, stats_tuple(st));
  py::list rl;
  for (int32_t r : rounds) rl.append(r);
  return py::make_tuple(betas, gn_obj, infos, kl, stats_tuple(st), rl);
}

py::tuple solve_lanes(uintptr_t ds, const py::list& specs, int64_t n, int64_t p, int64_t G, double tol, int max_iter, int check_every,
                      double L, uint32_t flags, bool want_gn, bool extrapolate, const py::object& alloc) {
  return solve_lanes_any(ds, specs, n, p, G, tol, max_iter, check_every, L, flags, want_gn, extrapolate, alloc, false);
}

py::tuple solve_lanes_reweighted(uintptr_t ds, const py::list& specs, int64_t n, int64_t p, int64_t G, double tol, int max_iter,
                                 int check_every, double L, uint32_t flags, bool want_gn, const py::object& alloc) {
  return solve_lanes_any(ds, specs, n, p, G, tol, max_iter, check_every, L, flags, want_gn, false, alloc, true);
}

// slm_solve_path_lanes: one path walked by `n_lanes` lanes.  Returns (betas, group norms or None, records as bytes, stats).
py::tuple solve_path_lanes(uintptr_t ds, const py::object& points_in, int64_t p, int64_t G, const py::object& a, const py::object& b,
                           const py::object& d, const py::object& beta0, int n_lanes, double tol, int max_iter, int check_every, double L,
                           uint32_t flags, bool want_gn, bool extrapolate, const py::object& alloc) {
  arr_d pts = arr_d::ensure(points_in);
  if (!pts || pts.size() % 3 != 0 || pts.size() == 0) throw py::value_error("points must be (K, 3)");
  const int64_t K = pts.size() / 3;
  std::vector<double> gam((size_t)K, 0.0);
  if (extrapolate) extrapolation(pts.data(), K, gam.data());
  std::vector<slm_path_point> cp((size_t)K);
  for (int64_t k = 0; k < K; ++k) cp[(size_t)k] = slm_path_point{pts.data()[3 * k], pts.data()[3 * k + 1], pts.data()[3 * k + 2], gam[(size_t)k]};
  Keep keep;
  slm_penalty pen{vector_arg(a, p, "a", keep), vector_arg(b, G, "b", keep), vector_arg(d, G, "d", keep)};
  const double* b0 = vector_arg(beta0, p, "beta0", keep);
  py::array betas = alloc(K * p).cast<py::array>();
  py::object gn_obj = py::none();
  double* gn_ptr = nullptr;
  if (want_gn) {
    py::array g = alloc(K * G).cast<py::array>();
    gn_ptr = static_cast<double*>(g.mutable_data());
    gn_obj = g;
  }
  py::array infos = info_bytes(K);
  std::memset(infos.mutable_data(), 0, sizeof(slm_point_info) * (size_t)K);
  slm_solve_opts opts{tol, max_iter, check_every, L, flags};
  slm_solve_stats st;
  std::memset(&st, 0, sizeof(st));
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = slm_solve_path_lanes(reinterpret_cast<slm_dataset*>(ds), &pen, cp.data(), (int32_t)K, n_lanes, &opts, b0,
                              static_cast<double*>(betas.mutable_data()), gn_ptr, static_cast<slm_point_info*>(infos.mutable_data()), &st);
  }
  check(rc);
  return py::make_tuple(betas, gn_obj, infos, stats_tuple(st));
}

// slm_dataset_create from numpy arrays of any layout (C- and F-contiguous matrices go as they are).  Returns the handle.
uintptr_t dataset_create(uintptr_t eng, const py::array& X_in, const py::object& y_in, const py::object& rw_in) {
  py::array X = X_in;
  if (X.ndim() != 2) throw py::value_error("X must be 2-D");
  const bool f64 = py::isinstance<py::array_t<double>>(X);
  const bool c_ok = (X.flags() & py::array::c_style) != 0, f_ok = (X.flags() & py::array::f_style) != 0;
  if (!f64 || !(c_ok || f_ok)) {
    X = arr_d::ensure(X_in);
    if (!X) throw py::value_error("X is not convertible to float64");
  }
  const int64_t n = X.shape(0), p = X.shape(1);
  const bool c_order = (X.flags() & py::array::c_style) != 0;
  Keep keep;
  const double* y = vector_arg(y_in, n, "y", keep);
  if (!y) throw py::value_error("y is None");
  const double* rw = vector_arg(rw_in, n, "row_weight", keep);
  slm_dataset* out = nullptr;
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = slm_dataset_create(reinterpret_cast<slm_engine*>(eng), static_cast<const double*>(X.data()), n, p, c_order ? p : 1, c_order ? 1 : n, y,
                            rw, &out);
  }
  check(rc);
  return reinterpret_cast<uintptr_t>(out);
}

}  // namespace

PYBIND11_MODULE(_slm_binding, m) {
  m.doc() = "compiled binding of the hot calls of libslm_hip.so (include/slm_engine.h)";
  m.def("set_error_types", [](py::object engine_error, py::object nonfinite_error) {
    g_engine_error = std::move(engine_error);
    g_nonfinite_error = std::move(nonfinite_error);
  });
  m.def("abi_version", []() { return slm_abi_version(); });
  m.def("info_record_bytes", []() { return (int)sizeof(slm_point_info); });
  m.def("solve_lanes", &solve_lanes);
  m.def("solve_lanes_reweighted", &solve_lanes_reweighted);
  m.def("solve_path_lanes", &solve_path_lanes);
  m.def("dataset_create", &dataset_create);
  m.def("path_extrapolation", [](const arr_d& pts) {
    if (pts.size() % 3 != 0) throw py::value_error("points must be (K, 3)");
    const int64_t K = pts.size() / 3;
    py::array_t<double> out((py::ssize_t)K);
    extrapolation(pts.data(), K, out.mutable_data());
    return out;
  });
  // (the module's globals are Python objects: dropped while the interpreter is still up)
  auto cleanup = []() {
    g_engine_error = py::object();
    g_nonfinite_error = py::object();
  };
  py::module_::import("atexit").attr("register")(py::cpp_function(cleanup));
}
