// diag.h -- the one door for test hooks and A/B switches.  NOT product surface: the product's knobs are the seven fields of
// hnsw_mi355x_options (include/hnsw_mi355x.h); everything here exists so that the test tiers can force a code path that the
// defaults would pick by themselves only on other inputs (the exact two-heap traversal, the latency variants, the hashed
// visited sets, spills and hand-backs, ...) and so that a measurement can be repeated with one mechanism switched off.
// Until round 4 these were 27 separate HNSW_MI355X_* environment variables, some read once and some per call; now:
//   * a list "name=value,name=value" -- hnsw_mi355x_options::diagnostics (process-wide, hnsw_mi355x_set_options) or, when
//     that is unset, the environment variable HNSW_MI355X_DIAG -- read on EVERY use, so a test that changes it between two
//     calls gets what it asked for;
//   * names (defaults in brackets; DESIGN.md 4.1 says what each is for):
//       lat [1]  novis [2]  novis_insert [1]  sorted_top [1]  shadow [1]  overlap [1]  mfma [1]  vis_hash [-1 = by graph size]
//       vis_hash_cap [0]  cand_cap [0]  spill_cap [-1]  link_plan [1]  concurrent_queries [1]  stream_queries [1]
//       xw_dry [1]  xw_stage [1]  trace [0]  lean [1]
#pragma once
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

namespace hnsw {

inline std::string &diag_override() { static std::string s; return s; }
inline bool &diag_override_set() { static bool b = false; return b; }
inline std::mutex &diag_mutex() { static std::mutex m; return m; }

inline void set_diag_string(const char *s) // nullptr: back to the environment variable
{
    std::lock_guard<std::mutex> lk(diag_mutex());
    diag_override_set() = s != nullptr;
    diag_override() = s ? s : "";
}

inline int diag(const char *name, int dflt)
{
    std::string held;
    const char *list;
    {
        std::lock_guard<std::mutex> lk(diag_mutex());
        if (diag_override_set()) { held = diag_override(); list = held.c_str(); }
        else list = std::getenv("HNSW_MI355X_DIAG");
    }
    if (!list || !*list) return dflt;
    const size_t n = std::strlen(name);
    for (const char *p = list; *p;) {
        while (*p == ',' || *p == ' ') ++p;
        if (std::strncmp(p, name, n) == 0 && p[n] == '=') return std::atoi(p + n + 1);
        while (*p && *p != ',') ++p;
    }
    return dflt;
}

} // namespace hnsw
