"""The reference test-suite's own inputs and scenarios, restated.

The reference's MSTest suite (src/HNSWIndex.Tests/) draws every vector from
`new Random(65537).NextSingle()` (Utils.cs:5,35-49) and normalises with `Utils.Normalize`
(Utils.cs:10-30).  Both are restated here bit for bit -- `System.Random(seed)` by the oracle's
generator (oracle/hnsw_oracle.c, pinned to public .NET known answers in test_oracle_dotnet.py) --
so that every assertion of that suite is evaluated on exactly the reference's inputs, by the CPU
oracle (tests/test_reference_inputs.py, CPU tier) and by the HIP path (GPU tier).

`scenarios()` lists one entry per reference test; `run(scenario, adapter)` plays it against an
index adapter (oracle or product) and returns the measured values; the reference's assertion for the
scenario is `check(name, values)`.  tests/golden/reference/fixture.json holds the values the oracle
produced when the fixture was made (tests/golden/make_reference_fixture.py); both tiers are held to
them, so oracle, product and fixture can only move together.
"""
import hashlib

import numpy as np

import oracle

SEED = 65537  # Utils.cs:5


def random_vectors(dim, count):
    """Utils.RandomVectors (Utils.cs:35-49): one `new Random(65537)`, NextSingle() per element, row by row."""
    return oracle.dotnet_random_single(SEED, dim * count).reshape(count, dim).copy()


def normalize(v):
    """Utils.Normalize (Utils.cs:10-30) in its float order: magnitude += x*x sequentially in float32
    (RyuJIT emits mulss/addss, no contraction), (float)Math.Sqrt((double)magnitude), 1f / that, x *= factor."""
    v = np.ascontiguousarray(v, dtype=np.float32)
    sq = v * v
    mag2 = np.add.accumulate(sq, axis=1, dtype=np.float32)[:, -1]
    mag = np.sqrt(mag2.astype(np.float64)).astype(np.float32)
    factor = (np.float32(1.0) / mag).astype(np.float32)
    return (v * factor[:, None]).astype(np.float32)


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:32]


def recall_at_1(adapter, vectors, ids):
    """Utils.Recall(index, vectors, vectors) (Utils.cs:54-70): k = 1, label == ground truth.  Labels are
    distinct vectors, so label equality is id equality."""
    got, _ = adapter.knn_query(vectors, 1)
    return float((got[:, 0] == ids).mean())


def components_per_layer(levels, layer_edges):
    """GraphNavigator.GetConnectedComponentCounts (GraphNavigator.cs:331-426): weakly connected
    components among the nodes of each layer (test-side union-find over the exported lists)."""
    out = []
    for layer, (counts, edges) in enumerate(layer_edges):
        nodes = np.nonzero(levels >= layer)[0]
        parent = {int(i): int(i) for i in nodes}

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a
        for i in nodes:
            for e in edges[i, :counts[i]]:
                ra, rb = find(int(i)), find(int(e))
                if ra != rb:
                    parent[ra] = rb
        out.append(len({find(int(i)) for i in nodes}))
    return out


# ------------------------------------------------------------------ adapters
class OracleAdapter:
    """The CPU restatement behind the scenario interface."""
    kind = "oracle"

    def __init__(self, dim, metric, **p):
        self.dim, self.metric, self.p = dim, metric, p
        self.ix = oracle.OracleIndex(dim, metric, **p)

    def add_each(self, x):      # a loop of HNSWIndex.Add(item) (HNSWIndex.cs:55-65)
        return self.ix.add(x)

    def add_list(self, x):      # HNSWIndex.Add(List) (HNSWIndex.cs:70-78): the product's deterministic schedule
        return self.ix.add_batched(x, 65536)

    def knn_query(self, q, k):
        return self.ix.knn_query(q, k)

    def knn_query_threads(self, q, k):
        return self.ix.knn_query(q, k, threads=4)

    def remove(self, ids):
        self.ix.remove(ids)

    def range_query(self, q, r):
        return self.ix.range_query(q, r)

    def graph_hash(self):
        return self.ix.graph_hash()

    def count(self):
        return self.ix.count

    def ids(self):
        return self.ix.active_ids()

    def levels(self):
        return self.ix.levels()

    def layer_edges(self, max_edges):
        lv = self.levels()
        out = []
        for layer in range(int(lv.max()) + 1):
            stride = 2 * max_edges + 2
            counts = np.full(lv.size, -1, np.int32)
            edges = np.zeros((lv.size, stride), np.int32)
            for i in np.nonzero(lv >= layer)[0]:
                e = self.ix.edges(int(i), layer)
                counts[i] = e.size
                edges[i, :e.size] = e
            out.append((counts, edges))
        return out

    def in_out_balanced(self, max_edges):
        """HNSWInfo (HNSWInfo.cs:31-43): AvgOutEdges == AvgInEdges on every layer -- and, stronger, the
        in-edge lists are exactly the transpose of the out-edge lists."""
        lv = self.levels()
        live = set(self.ids().tolist())
        for layer in range(int(lv.max()) + 1):
            n_out = n_in = 0
            fwd, back = set(), set()
            for i in live:
                if lv[i] < layer:
                    continue
                o = self.ix.edges(i, layer).tolist()
                b = self.ix.edges(i, layer, incoming=True).tolist()
                n_out += len(o)
                n_in += len(b)
                fwd.update((i, t) for t in o)
                back.update((s, i) for s in b)
            if n_out != n_in or fwd != back:
                return False
        return True

    def max_in_edges(self):
        lv = self.levels()
        return max((self.ix.edges(int(i), l, incoming=True).size for i in range(lv.size) for l in range(lv[i] + 1)), default=0)


class ProductAdapter:
    """The HIP path through the C ABI (hnswindex.Index)."""
    kind = "product"

    def __init__(self, dim, metric, **p):
        import hnswindex
        self.dim, self.metric, self.p = dim, metric, p
        self.ix = hnswindex.Index(dim, metric)
        self._sequential = None

    def _create(self, sequential):
        if self.ix._initialized:
            assert self._sequential == sequential, "one Add style per scenario"
            return
        p, ix = self.p, self.ix
        ix.set_max_edges(p.get("max_edges", 16))
        ix.set_min_nn(p.get("min_nn", 5))
        ix.set_max_candidates(p.get("max_candidates", 100))
        ix.set_collection_size(p.get("collection_size", 65536))
        ix.set_random_seed(p.get("random_seed", 31337))
        ix.set_allow_removals(p.get("allow_removals", True))
        ix.set_insert_batch(1 if sequential else 65536)
        self._sequential = sequential

    def add_each(self, x):      # count == 1 batches: exactly the sequential HNSWIndex.Add(item)
        self._create(True)
        return self.ix.add(x)

    def add_list(self, x):
        self._create(False)
        return self.ix.add(x)

    def knn_query(self, q, k):
        return self.ix.knn_query(q, k)

    def knn_query_threads(self, q, k):
        """QueryGraphMultiThread (GraphTests.cs:82-120): the same queries from several host threads at once,
        on one handle (ctypes drops the GIL inside the call)."""
        import threading
        q = np.ascontiguousarray(q, dtype=np.float32)
        parts = np.array_split(np.arange(q.shape[0]), 4)
        res = [None] * len(parts)

        def work(t):
            ids, d = [], []
            for lo in range(0, parts[t].size, 97):   # many overlapping calls per thread
                a, b = self.ix.knn_query(q[parts[t][lo:lo + 97]], k)
                ids.append(a); d.append(b)
            res[t] = (np.concatenate(ids), np.concatenate(d))
        th = [threading.Thread(target=work, args=(t,)) for t in range(len(parts))]
        [t.start() for t in th]
        [t.join() for t in th]
        assert all(r is not None for r in res)
        return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])

    def remove(self, ids):
        self.ix.remove(ids)

    def range_query(self, q, r):
        return self.ix.range_query(q, r)

    def graph_hash(self):
        return self.ix.graph_hash()

    def count(self):
        return self.ix.count

    def ids(self):
        return self.ix.ids()

    def levels(self):
        return self.ix.levels()

    def layer_edges(self, max_edges):
        lv = self.levels()
        return [self.ix.export_edges(l, 2 * max_edges + 2) for l in range(int(lv.max()) + 1)]

    def in_out_balanced(self, max_edges):
        """The product keeps no in-edge lists during Add (rebuilt by transposition when a
        removal needs them), so the reference's AvgOut == AvgIn reduces to: every out-edge of a live node
        names a live node that has that layer (its transpose is then a valid in-edge list)."""
        lv = self.levels()
        live = np.zeros(lv.size, bool)
        live[self.ids()] = True
        for layer, (counts, edges) in enumerate(self.layer_edges(max_edges)):
            for i in np.nonzero(live & (lv >= layer))[0]:
                t = edges[i, :counts[i]]
                if not (live[t].all() and (lv[t] >= layer).all()):
                    return False
        return True

    def max_in_edges(self):
        return 0  # none are kept (see in_out_balanced)


# ------------------------------------------------------------------ scenarios
def scenarios():
    """name -> (reference test, runner).  Runners take an adapter factory `mk(dim, metric, **params)`."""
    S = {}

    def build_unit(mk, n=2000, adder="add_each", **p):
        v = normalize(random_vectors(128, n))
        a = mk(128, "ucosine", **p)
        ids = getattr(a, adder)(v)
        return a, v, ids

    def graph_single_thread(mk):      # GraphTests.BuildGraphSingleThread :16-37
        a, v, ids = build_unit(mk)
        qi, qd = a.knn_query(v, 10)
        ti, td = a.knn_query_threads(v, 10)   # QueryGraphMultiThread :82-120 on the same index
        return dict(recall=recall_at_1(a, v, ids), balanced=a.in_out_balanced(16), graph_hash=str(a.graph_hash()),
                    levels=sha(a.levels()), knn10=sha(qi, qd), threads_equal=bool((qi == ti).all() and qd.tobytes() == td.tobytes()))
    S["GraphTests.BuildGraphSingleThread+QueryGraphMultiThread"] = graph_single_thread

    def graph_multi_thread(mk):       # GraphTests.BuildGraphMultiThread :39-59 (Parallel.For of Add(item))
        a, v, ids = build_unit(mk, adder="add_list")
        return dict(recall=recall_at_1(a, v, ids), balanced=a.in_out_balanced(16), graph_hash=str(a.graph_hash()))
    S["GraphTests.BuildGraphMultiThread"] = graph_multi_thread

    def graph_batch(mk):              # GraphTests.BuildGraphBatch :61-80: cosine, NOT normalised, Add(List)
        v = random_vectors(128, 2000)
        a = mk(128, "cosine")
        ids = a.add_list(v)
        return dict(recall=recall_at_1(a, v, ids), balanced=a.in_out_balanced(16), graph_hash=str(a.graph_hash()))
    S["GraphTests.BuildGraphBatch"] = graph_batch

    def remove_nodes(mk):             # GraphTests.RemoveNodesTest / ...Parallel / ...Batch :122-225
        a, v, ids = build_unit(mk)
        insert_recall = recall_at_1(a, v, ids)
        odd, even = ids[1::2], ids[0::2]
        a.remove(odd)                 # ids in order; the reference's three variants differ only in scheduling
        removal_recall = recall_at_1(a, v[0::2], even)
        return dict(insert_recall=insert_recall, removal_recall=removal_recall, balanced=a.in_out_balanced(16),
                    graph_hash=str(a.graph_hash()), count=int(a.count()))
    S["GraphTests.RemoveNodesTest"] = remove_nodes

    def range_query(mk):              # GraphTests.RangeQueryTest :227-244
        v = random_vectors(128, 2000)
        a = mk(128, "sq_euclid")
        a.add_each(v)
        ids, d = a.range_query(v, 32.0)
        return dict(all_within=bool(all((x <= 32.0).all() for x in d)), counts=sha(np.array([x.size for x in ids], np.int64)),
                    total=int(sum(x.size for x in ids)), results=sha(np.concatenate(ids), np.concatenate(d)))
    S["GraphTests.RangeQueryTest"] = range_query

    def components(mk):               # GraphTests.ConnectedComponentCountsPerLayerTest :253-273
        v = normalize(random_vectors(128, 2000)[:256])
        a = mk(128, "ucosine", random_seed=12345)
        a.add_each(v)
        comp = components_per_layer(a.levels(), a.layer_edges(16))
        return dict(components=comp, graph_hash=str(a.graph_hash()), levels=sha(a.levels()))
    S["GraphTests.ConnectedComponentCountsPerLayerTest"] = components

    def params(mk, metric, normalise, adder, **p):
        v = random_vectors(128, 1000)
        if normalise:
            v = normalize(v)
        a = mk(128, metric, **p)
        ids = getattr(a, adder)(v)
        return a, v, ids

    def p_min_nn(mk):                 # ParametersTests.TestParameterMinNN :14-30
        a, v, ids = params(mk, "cosine", True, "add_each", min_nn=1)
        return dict(recall=recall_at_1(a, v, ids), graph_hash=str(a.graph_hash()))
    S["ParametersTests.TestParameterMinNN"] = p_min_nn

    def p_max_candidates(mk):         # :32-48
        a, v, ids = params(mk, "cosine", True, "add_each", max_candidates=32)
        return dict(recall=recall_at_1(a, v, ids), graph_hash=str(a.graph_hash()))
    S["ParametersTests.TestParameterMaxCandidates"] = p_max_candidates

    def p_low_recall(mk):             # :50-66
        a, v, ids = params(mk, "cosine", True, "add_each", max_edges=8, min_nn=1, max_candidates=16)
        return dict(recall=recall_at_1(a, v, ids), graph_hash=str(a.graph_hash()))
    S["ParametersTests.TestParameterLowRecall"] = p_low_recall

    def p_allow_removals(mk):         # :68-88
        a, v, ids = params(mk, "sq_euclid", False, "add_list", allow_removals=False)
        try:
            a.remove([0])
            threw = False
        except RuntimeError:
            threw = True
        return dict(recall=recall_at_1(a, v, ids), max_in_edges=int(a.max_in_edges()), remove_throws=threw, graph_hash=str(a.graph_hash()))
    S["ParametersTests.TestParameterAllowRemovals"] = p_allow_removals

    def resize(mk, adder):            # GraphResizeTests.SingleThreadGraphResize / MultiThreadGraphResize :16-60
        v = normalize(random_vectors(128, 5000))
        a = mk(128, "sq_euclid", collection_size=10)
        ids = getattr(a, adder)(v)
        return dict(recall=recall_at_1(a, v, ids), balanced=a.in_out_balanced(16), graph_hash=str(a.graph_hash()))
    S["GraphResizeTests.SingleThreadGraphResize"] = lambda mk: resize(mk, "add_each")
    S["GraphResizeTests.MultiThreadGraphResize"] = lambda mk: resize(mk, "add_list")

    def active_set(mk):               # GraphResizeTests.ActiveSetContainsCorrectTest :62-88
        v = normalize(random_vectors(128, 5000))
        a = mk(128, "sq_euclid")
        ids = a.add_each(v)
        a.remove(ids[1::2])
        live = set(a.ids().tolist())
        return dict(count=int(a.count()), only_even=bool(live == set(ids[0::2].tolist())), graph_hash=str(a.graph_hash()))
    S["GraphResizeTests.ActiveSetContainsCorrectTest"] = active_set

    def remove_all(mk):               # GraphResizeTests.RemoveAllTest :90-104 (RemoveAllParallelTest :106-120: same end state)
        v = random_vectors(128, 5000)
        a = mk(128, "sq_euclid", collection_size=10)
        a.add_each(v)
        ok = True
        for i in range(5000):
            a.remove([i])
            ok = ok and a.count() == 5000 - i - 1
        return dict(count_after_each=bool(ok), final_count=int(a.count()))
    S["GraphResizeTests.RemoveAllTest"] = remove_all
    return S


def check(name, r):
    """The reference's own assertion for each scenario (file:line in scenarios())."""
    if name.startswith("GraphTests.BuildGraph"):
        assert r["recall"] > 0.85 and r["balanced"]
        if "threads_equal" in r:
            assert r["threads_equal"]
    elif name == "GraphTests.RemoveNodesTest":
        assert r["insert_recall"] * 0.98 < r["removal_recall"] and r["balanced"] and r["count"] == 1000
    elif name == "GraphTests.RangeQueryTest":
        assert r["all_within"]
    elif name == "GraphTests.ConnectedComponentCountsPerLayerTest":
        assert len(r["components"]) >= 1 and all(c == 1 for c in r["components"])
    elif name == "ParametersTests.TestParameterMinNN":
        assert 0.70 < r["recall"] < 0.90
    elif name == "ParametersTests.TestParameterMaxCandidates":
        assert r["recall"] > 0.90
    elif name == "ParametersTests.TestParameterLowRecall":
        assert r["recall"] < 0.50
    elif name == "ParametersTests.TestParameterAllowRemovals":
        assert r["recall"] > 0.9 and r["max_in_edges"] == 0 and r["remove_throws"]
    elif name.startswith("GraphResizeTests.") and "Resize" in name.split(".")[1]:
        assert r["recall"] > 0.85 and r["balanced"]
    elif name == "GraphResizeTests.ActiveSetContainsCorrectTest":
        assert r["count"] == 2500 and r["only_even"]
    elif name == "GraphResizeTests.RemoveAllTest":
        assert r["count_after_each"] and r["final_count"] == 0
    else:
        raise KeyError(name)
