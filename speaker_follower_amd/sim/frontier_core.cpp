// Bookkeeping of the state-factored search as a native object: one call per search
// iteration between two decoder steps on the device.
//
// What it computes is tasks/R2R/follower.py:720-980 (`Seq2SeqAgent.state_factored_search`) in the form of
// speaker_follower_amd/frontier.py: a hypothesis is a ROW of integer / float32 arrays with a parent pointer, a
// world state is an integer into the navigation tables (state id = nav row * 36 + view), env.step is the table
// look-up next_row[s, a] / cand_view[s, a].  The reference keeps, per instance, two dictionaries {state key:
// hypothesis} (states to expand -- `cache`, follower.py:757 -- and finished hypotheses waiting for their turn --
// `completed_holding`, :758), walks every successor of the expanded states in descending score order keeping, per
// key, a successor only if it beats what the dictionary holds (:842-856), and then expands the best entries not
// expanded yet (heapq.nlargest over chain(cache, holding), :861-865: ties go to states-to-expand first, then to
// the older entry).  Here each dictionary is a per-instance hash map key -> slot plus slot-ordered arrays (slots
// are handed out in insertion order, so "first maximum" IS the reference's tie order), and one iteration costs
// microseconds instead of the ~100 numpy calls of frontier._state_factored_search_numpy, which stays as the
// readable restatement and is compared with this object result for result in tests/test_search_host.py.
// Scores are accumulated in float32, every partial sum rounded (follower.py:826: a float32 tensor element is
// added), as on the numpy path.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <unordered_map>
#include <vector>

namespace py = pybind11;
using i64 = int64_t;

namespace {

constexpr float kNegInf = -std::numeric_limits<float>::infinity();

template <typename T>
using Arr = py::array_t<T, py::array::c_style | py::array::forcecast>;

struct Dict {                               // one per-instance dictionary of the reference
    std::unordered_map<i64, int> slot_of;   // key -> slot
    std::vector<i64> key, node;             // per slot, insertion order
    std::vector<float> pick;                // score while waiting to be expanded, -inf once expanded

    void put(i64 k, i64 n, float score) {
        auto it = slot_of.find(k);
        if (it == slot_of.end()) {
            slot_of.emplace(k, (int)key.size());
            key.push_back(k);
            node.push_back(n);
            pick.push_back(score);
        } else {
            node[it->second] = n;
            pick[it->second] = score;
        }
    }
    // first maximum of `pick` (the oldest entry among equals); -1 if nothing waits
    int best(float* value) const {
        int arg = -1;
        float m = kNegInf;
        for (int i = 0; i < (int)pick.size(); ++i)
            if (pick[i] > m) {
                m = pick[i];
                arg = i;
            }
        *value = m;
        return arg;
    }
};

struct Instance {
    Dict open, held;
    std::unordered_map<i64, i64> completed;     // end state key -> hypothesis
    std::vector<i64> completed_order;           // keys in the order they were first completed
    std::vector<i64> visits;                    // successively expanded hypotheses
    int n_completed = 0;
};

struct Succ {
    float score;
    int owner, action;
};

class StateFactored {
  public:
    StateFactored(int completion_size, int successor_size, int episode_len, int key_fields, int V, Arr<int32_t> next_row,
                  Arr<int32_t> cand_view, Arr<int32_t> a_num, Arr<i64> base_row, Arr<i64> root_sid, Arr<i64> root_key)
        : completion_(completion_size), successor_(successor_size), episode_len_(episode_len), key_fields_(key_fields),
          V_(V), next_row_(next_row), cand_view_(cand_view), a_num_(a_num) {
        if (next_row.ndim() != 2 || cand_view.ndim() != 2 || a_num.ndim() != 1)
            throw std::invalid_argument("next_row / cand_view [states, A], a_num [states]");
        A_ = (int)next_row.shape(1);
        nr_ = next_row_.data();
        cv_ = cand_view_.data();
        an_ = a_num_.data();
        const int B = (int)root_sid.shape(0);
        base_row_.assign(base_row.data(), base_row.data() + B);
        inst_.resize(B);
        for (int b = 0; b < B; ++b) {
            const i64 id = append(-1, b, root_sid.data()[b], root_key.data()[b], -1, 0, b, 0.f, true);
            inst_[b].open.put(root_key.data()[b], id, kNegInf);          // the roots: expanded from the start
            inst_[b].visits.push_back(id);
            frontier_.push_back(id);
        }
    }

    int n_frontier() const { return (int)frontier_.size(); }
    i64 n_hypotheses() const { return (i64)parent_.size(); }
    bool done() const {
        for (const auto& in : inst_)
            if (in.n_completed < completion_) return false;
        return true;
    }

    // Index-form inputs of the decoder step over the current frontier: rows of `out` [8, cap] = nav row, view,
    // parent's nav row, parent's view, action that led here (0 for a root: the zero embedding, model.py:368), pool
    // row of (h, c), instruction row, pool row the new state is written to (base + i).  Columns beyond the frontier
    // repeat column 0 with destination -1.
    int fill_inputs(py::array_t<int32_t, py::array::c_style> out, i64 base) const {
        if (out.ndim() != 2 || out.shape(0) != 8) throw std::invalid_argument("inputs buffer must be int32 [8, cap]");
        return fill_inputs_raw(out.mutable_data(), (int)out.shape(1), base);
    }
    int fill_inputs_raw(int32_t* o, int cap, i64 base) const {
        const int N = (int)frontier_.size();
        if (N > cap || N == 0) throw std::invalid_argument("frontier does not fit the inputs buffer");
        for (int i = 0; i < cap; ++i) {
            const i64 f = frontier_[i < N ? i : 0];
            const i64 s = sid_[f], p = parent_[f];
            const i64 ps = p >= 0 ? sid_[p] : s;
            // (rows 0-1: nav rows of the states, then of their parents; rows 2-3: their views likewise -- one
            // navigation look-up over 2 cap entries serves both)
            o[0 * cap + i] = (int32_t)(s / V_);
            o[1 * cap + i] = (int32_t)(ps / V_);
            o[2 * cap + i] = (int32_t)(s % V_);
            o[3 * cap + i] = (int32_t)(ps % V_);
            o[4 * cap + i] = p >= 0 ? (int32_t)action_[f] : 0;
            o[5 * cap + i] = (int32_t)pool_[f];
            o[6 * cap + i] = (int32_t)hinst_[f];
            o[7 * cap + i] = i < N ? (int32_t)(base + i) : -1;
        }
        return N;
    }

    // One iteration (follower.py:783-905) given log_softmax of the decoder step over the current frontier: logp
    // [>= n_frontier, >= max a_num] float32, row i = frontier state i, new states' pool rows = base + i.  Returns
    // the size of the next frontier.
    int advance(py::array_t<float, py::array::c_style> logp, i64 base) {
        if (logp.ndim() != 2) throw std::invalid_argument("logp must be float32 [states, actions]");
        if (logp.shape(0) < (i64)frontier_.size()) throw std::invalid_argument("logp has fewer rows than the frontier");
        return advance_raw(logp.data(), logp.shape(1), base);
    }
    // The whole loop of the search natively (follower.py:783-905 iterated): inputs of the frontier into the pinned
    // block -> launch of the decoder-step graph -> stream sync -> bookkeeping, until every instance has its completions,
    // the frontier is empty, or the next states would not fit the state pool (the caller grows it and calls again).
    // `launch` / `sync` are hipGraphLaunch / hipStreamSynchronize of the process's HIP runtime, handed over as
    // addresses (this module does not link against HIP); the GIL is released while it runs.
    // Returns (status, base, iterations): 0 finished, 1 pool full, < 0: -(hip error of the launch / sync).
    py::tuple run_graph(std::uintptr_t launch, std::uintptr_t sync, std::uintptr_t graph_exec, std::uintptr_t stream,
                        py::array_t<int32_t, py::array::c_style> inputs, py::array_t<float, py::array::c_style> logp,
                        i64 base, i64 pool_rows) {
        if (inputs.ndim() != 2 || inputs.shape(0) != 8) throw std::invalid_argument("inputs buffer must be int32 [8, cap]");
        if (logp.ndim() != 2 || logp.shape(0) < inputs.shape(1)) throw std::invalid_argument("logp must be float32 [>= cap, actions]");
        using LaunchFn = int (*)(void*, void*);
        using SyncFn = int (*)(void*);
        const LaunchFn do_launch = reinterpret_cast<LaunchFn>(launch);
        const SyncFn do_sync = reinterpret_cast<SyncFn>(sync);
        int32_t* in = inputs.mutable_data();
        const int cap = (int)inputs.shape(1);
        const float* lp = logp.data();
        const i64 ld = logp.shape(1);
        int status = 0, iterations = 0;
        {
            py::gil_scoped_release nogil;
            for (;;) {
                const int n = (int)frontier_.size();
                if (n == 0 || done()) break;
                if (base + n > pool_rows) { status = 1; break; }
                fill_inputs_raw(in, cap, base);
                int rc = do_launch(reinterpret_cast<void*>(graph_exec), reinterpret_cast<void*>(stream));
                if (rc == 0) rc = do_sync(reinterpret_cast<void*>(stream));
                if (rc != 0) { status = -rc; break; }
                ++iterations;
                const int next = advance_raw(lp, ld, base);
                base += n;
                if (next == 0) break;
            }
        }
        return py::make_tuple(status, base, iterations);
    }
    int advance_raw(const float* lp, const i64 ld, i64 base) {
        const int N = (int)frontier_.size();
        // ---- successors of every expanded state, instance by instance (the frontier is grouped by instance)
        int i = 0;
        while (i < N) {
            const int b = (int)hinst_[frontier_[i]];
            int j = i;
            while (j < N && hinst_[frontier_[j]] == b) ++j;
            Instance& in = inst_[b];
            if (in.n_completed < completion_) {
                succ_.clear();
                for (int r = i; r < j; ++r) {
                    const i64 f = frontier_[r];
                    const int na = an_[sid_[f]];
                    if (na > ld) throw std::invalid_argument("logp has fewer columns than a state has candidates");
                    for (int a = 0; a < na; ++a) succ_.push_back({score_[f] + lp[(i64)r * ld + a], r, a});
                }
                // descending score, generation order among equals (follower.py:842: sorted(..., reverse) is stable)
                std::stable_sort(succ_.begin(), succ_.end(), [](const Succ& x, const Succ& y) { return x.score > y.score; });
                seen_.clear();
                for (const Succ& s : succ_) {
                    const i64 f = frontier_[s.owner];
                    const i64 st = sid_[f];
                    const i64 nxt = nr_[st * A_ + s.action];
                    const bool stay = s.action == 0 || nxt == st / V_;      // env.py:126-146
                    const i64 nsid = stay ? st : nxt * V_ + cv_[st * A_ + s.action];
                    const i64 key = stay ? key_[f] : key_of(nsid, b);
                    const i64 count = count_[f] + 1;
                    const bool final = s.action == 0 || count == episode_len_;
                    const i64 group = key * 2 + (final ? 1 : 0);
                    if (std::find(seen_.begin(), seen_.end(), group) != seen_.end()) continue;   // a better one came first
                    seen_.push_back(group);
                    Dict& d = final ? in.held : in.open;
                    auto it = d.slot_of.find(key);
                    if (it != d.slot_of.end() && !(score_[d.node[it->second]] < s.score)) continue;
                    const i64 id = append(f, b, nsid, key, s.action, count, base + s.owner, s.score, start_pose_[f] && stay);
                    d.put(key, id, s.score);
                }
            }
            i = j;
        }
        // ---- per instance: the `successor_size` best entries not expanded yet
        frontier_.clear();
        for (int b = 0; b < (int)inst_.size(); ++b) {
            Instance& in = inst_[b];
            if (in.n_completed >= completion_) continue;
            picks_.clear();
            for (int r = 0; r < successor_; ++r) {
                float mo, mh;
                const int ao = in.open.best(&mo), ah = in.held.best(&mh);
                if (ao < 0 && ah < 0) break;
                const bool use_h = mh > mo;                                  // equal scores: the state to expand first
                Dict& d = use_h ? in.held : in.open;
                const int slot = use_h ? ah : ao;
                d.pick[slot] = kNegInf;                                      // expanded
                picks_.push_back({d.key[slot], d.node[slot], use_h});
            }
            // finished hypotheses: their end state is completed (the better one if it already was)
            for (const Pick& p : picks_) {
                if (!p.held) continue;
                auto it = in.completed.find(p.key);
                if (it == in.completed.end()) {
                    in.completed.emplace(p.key, p.node);
                    in.completed_order.push_back(p.key);
                    ++in.n_completed;
                } else if (score_[it->second] < score_[p.node]) {
                    it->second = p.node;
                }
            }
            // states to expand next; an instance that has reached its completions stops
            for (const Pick& p : picks_)
                if (!p.held && in.n_completed < completion_) {
                    frontier_.push_back(p.node);
                    in.visits.push_back(p.node);
                }
        }
        return (int)frontier_.size();
    }

    // (parent, inst, sid, key, action, count, pool) int64 [n], score float32 [n], start_pose bool [n]
    py::tuple hypotheses() const {
        return py::make_tuple(vec(parent_), vec(hinst_), vec(sid_), vec(key_), vec(action_), vec(count_), vec(pool_),
                              vec(score_), vec(start_pose_));
    }

    // per instance: its completed end states' hypotheses, best first (stable), at most completion_size; and the
    // successively expanded hypotheses followed by those (follower.py:907-925)
    py::tuple results() const {
        py::list completed, visits;
        for (const Instance& in : inst_) {
            std::vector<i64> nodes;
            for (i64 k : in.completed_order) nodes.push_back(in.completed.at(k));
            std::stable_sort(nodes.begin(), nodes.end(), [this](i64 x, i64 y) { return score_[x] > score_[y]; });
            if ((int)nodes.size() > completion_) nodes.resize(completion_);
            std::vector<i64> v = in.visits;
            v.insert(v.end(), nodes.begin(), nodes.end());
            completed.append(nodes);
            visits.append(v);
        }
        return py::make_tuple(completed, visits);
    }

  private:
    struct Pick {
        i64 key, node;
        bool held;
    };

    template <typename T>
    static py::array_t<T> vec(const std::vector<T>& v) {
        return py::array_t<T>((py::ssize_t)v.size(), v.data());
    }

    i64 key_of(i64 sid, int b) const {                                       // follower.py:722, 843: world_state[0:n]
        const i64 local_row = sid / V_ - base_row_[b];
        if (key_fields_ == 4) return local_row * V_ + sid % V_;
        if (key_fields_ == 3) return local_row * 12 + sid % 12;
        if (key_fields_ == 2) return local_row;
        return 0;
    }

    i64 append(i64 parent, i64 inst, i64 sid, i64 key, i64 action, i64 count, i64 pool, float score, bool start_pose) {
        parent_.push_back(parent);
        hinst_.push_back(inst);
        sid_.push_back(sid);
        key_.push_back(key);
        action_.push_back(action);
        count_.push_back(count);
        pool_.push_back(pool);
        score_.push_back(score);
        start_pose_.push_back(start_pose ? 1 : 0);
        return (i64)parent_.size() - 1;
    }

    int completion_, successor_, episode_len_, key_fields_, V_, A_ = 0;
    Arr<int32_t> next_row_, cand_view_, a_num_;          // (kept alive: the raw pointers below point into them)
    const int32_t *nr_ = nullptr, *cv_ = nullptr, *an_ = nullptr;
    std::vector<i64> base_row_;
    std::vector<Instance> inst_;
    std::vector<i64> frontier_;
    std::vector<i64> parent_, hinst_, sid_, key_, action_, count_, pool_;
    std::vector<float> score_;
    std::vector<uint8_t> start_pose_;
    std::vector<Succ> succ_;
    std::vector<i64> seen_;
    std::vector<Pick> picks_;
};

}  // namespace

PYBIND11_MODULE(sf_frontier, m) {
    m.doc() = "state-factored search bookkeeping (follower.py:720-980) over integer world states";
    py::class_<StateFactored>(m, "StateFactored")
        .def(py::init<int, int, int, int, int, Arr<int32_t>, Arr<int32_t>, Arr<int32_t>, Arr<i64>, Arr<i64>, Arr<i64>>(),
             py::arg("completion_size"), py::arg("successor_size"), py::arg("episode_len"), py::arg("key_fields"),
             py::arg("n_views"), py::arg("next_row"), py::arg("cand_view"), py::arg("a_num"), py::arg("base_row"),
             py::arg("root_sid"), py::arg("root_key"))
        .def_property_readonly("n_frontier", &StateFactored::n_frontier)
        .def_property_readonly("n_hypotheses", &StateFactored::n_hypotheses)
        .def("done", &StateFactored::done)
        .def("fill_inputs", &StateFactored::fill_inputs, py::arg("out"), py::arg("base"))
        .def("advance", &StateFactored::advance, py::arg("logp"), py::arg("base"))
        .def("run_graph", &StateFactored::run_graph, py::arg("launch"), py::arg("sync"), py::arg("graph_exec"), py::arg("stream"),
             py::arg("inputs"), py::arg("logp"), py::arg("base"), py::arg("pool_rows"))
        .def("hypotheses", &StateFactored::hypotheses)
        .def("results", &StateFactored::results);
}
