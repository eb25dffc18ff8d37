// tools/dotnet_fixture/Program.cs -- produces fixtures from the REAL reference (Skaipi/HNSWIndex.Net, C#) for this repo's parity
// tests.  NOT compiled in this repo's environment (no dotnet SDK there): a maintainer with the SDK and a checkout of the
// reference runs it once and commits the output under tests/golden/dotnet/ -- from then on the oracle (CPU tier) and the HIP
// path (GPU tier) are held to genuine .NET output (tests/test_dotnet_fixtures.py), which is what turns "parity unpinned" for
// System.Random's sample stream, NextSingle's redraw, Span.Sort's order among equal keys, the heaps' tie order and
// protobuf-net's wire choices into pinned.
//
//   python tools/dotnet_fixture/export_inputs.py            # writes tools/dotnet_fixture/inputs/{cases.json,*.f32}
//   cd tools/dotnet_fixture && dotnet run -c Release -- inputs ../../tests/golden/dotnet
//
// For every case of inputs/cases.json (the seeds / shapes / parameters of tests/golden/*.json, a tie-heavy integer-grid case
// and a removal case) it does what the reference's own determinism recipe does (bindings/__tests__/parameters_test.py:65-68):
// one HNSWIndex.Add(item) per vector, in order (src/HNSWIndex/HNSWIndex.cs:55-65); then KnnQuery per query (:107-124), optional
// Remove (:83-92) + queries again, and Serialize (:210-219).  Written per case: <name>.json (ids returned by Add, knn ids and
// the distances' IEEE bit patterns, range-query results, Count) and <name>.snapshot (protobuf-net's bytes, from which the
// tests read levels and adjacency lists).  Only the reference's PUBLIC API is used.
using System.Globalization;
using System.Text;
using System.Text.Json;
using HNSWIndex;

static class Program
{
    sealed class Case
    {
        public string name { get; set; } = "";
        public int n { get; set; }
        public int dim { get; set; }
        public string metric { get; set; } = "sq_euclid";      // sq_euclid | cosine | ucosine   (HNSWIndexExports.cs:47-60)
        public int nq { get; set; }
        public int k { get; set; }
        public int max_edges { get; set; } = 16;
        public int max_candidates { get; set; } = 100;
        public int min_nn { get; set; } = 5;
        public int collection_size { get; set; } = 65536;
        public int random_seed { get; set; } = 31337;
        public bool allow_removals { get; set; } = true;
        public int[] remove { get; set; } = Array.Empty<int>(); // ids removed (in this order) after the first round of queries
        public float range { get; set; } = -1f;                // >= 0: also RangeQuery with this radius
    }

    static float[][] ReadRows(string path, int n, int dim)
    {
        var bytes = File.ReadAllBytes(path);
        if (bytes.Length != 4L * n * dim) throw new InvalidDataException($"{path}: {bytes.Length} bytes, expected {4L * n * dim}");
        var rows = new float[n][];
        for (int i = 0; i < n; i++)
        {
            rows[i] = new float[dim];
            Buffer.BlockCopy(bytes, 4 * i * dim, rows[i], 0, 4 * dim);   // little-endian float32, as numpy wrote them
        }
        return rows;
    }

    static Func<float[], float[], float> Metric(string name) => name switch
    {
        "sq_euclid" => SquaredEuclideanMetric.Compute,   // src/HNSWIndex/Metrics/EuclideanMetric.cs:11
        "cosine" => CosineMetric.Compute,                // src/HNSWIndex/Metrics/CosineMetric.cs:10
        "ucosine" => CosineMetric.UnitCompute,           // :95
        _ => throw new ArgumentException("Unsupported distance metric: " + name),
    };

    static void WriteResults(Utf8JsonWriter w, string key, List<KNNResult<float[], float>>[] res)
    {
        w.WriteStartObject(key);
        w.WriteStartArray("ids");
        foreach (var r in res) { w.WriteStartArray(); foreach (var e in r) w.WriteNumberValue(e.Id); w.WriteEndArray(); }
        w.WriteEndArray();
        w.WriteStartArray("dist_bits");                  // BitConverter.SingleToUInt32Bits: the exact float, no printing round trip
        foreach (var r in res) { w.WriteStartArray(); foreach (var e in r) w.WriteNumberValue(BitConverter.SingleToUInt32Bits(e.Distance)); w.WriteEndArray(); }
        w.WriteEndArray();
        w.WriteEndObject();
    }

    static int Main(string[] args)
    {
        if (args.Length < 2) { Console.Error.WriteLine("usage: dotnet run -c Release -- <inputs dir> <output dir>"); return 64; }
        string inDir = args[0], outDir = args[1];
        Directory.CreateDirectory(outDir);
        var cases = JsonSerializer.Deserialize<List<Case>>(File.ReadAllText(Path.Combine(inDir, "cases.json")))!;
        foreach (var c in cases)
        {
            var x = ReadRows(Path.Combine(inDir, c.name + ".x.f32"), c.n, c.dim);
            var q = ReadRows(Path.Combine(inDir, c.name + ".q.f32"), c.nq, c.dim);
            var p = new HNSWParameters<float>
            {
                MaxEdges = c.max_edges, MaxCandidates = c.max_candidates, MinNN = c.min_nn, CollectionSize = c.collection_size,
                RandomSeed = c.random_seed, AllowRemovals = c.allow_removals,   // DistributionRate, RemoveMaxCandidates: defaults (HNSWParameters.cs:19,37)
            };
            var index = new HNSWIndex<float[], float>(Metric(c.metric), p);
            var addIds = new int[c.n];
            for (int i = 0; i < c.n; i++) addIds[i] = index.Add(x[i]);          // one at a time: the only Add whose graph is defined
            var knn = new List<KNNResult<float[], float>>[c.nq];
            for (int i = 0; i < c.nq; i++) knn[i] = index.KnnQuery(q[i], c.k);   // single-threaded, in order
            index.Serialize(Path.Combine(outDir, c.name + ".snapshot"));
            using var fs = File.Create(Path.Combine(outDir, c.name + ".json"));
            using var w = new Utf8JsonWriter(fs);
            w.WriteStartObject();
            w.WriteString("name", c.name);
            w.WriteString("produced_by", "Skaipi/HNSWIndex.Net, " + System.Runtime.InteropServices.RuntimeInformation.FrameworkDescription);
            w.WriteBoolean("avx_fma", System.Runtime.Intrinsics.X86.Avx.IsSupported && System.Runtime.Intrinsics.X86.Fma.IsSupported); // the branch the metrics took (EuclideanMetric.cs:19)
            w.WriteStartArray("add_ids"); foreach (var id in addIds) w.WriteNumberValue(id); w.WriteEndArray();
            w.WriteNumber("count", index.Count);
            WriteResults(w, "knn", knn);
            if (c.range >= 0)
            {
                var rr = new List<KNNResult<float[], float>>[c.nq];
                for (int i = 0; i < c.nq; i++) rr[i] = index.RangeQuery(q[i], c.range);   // HNSWIndex.cs:144-158
                WriteResults(w, "range", rr);
            }
            if (c.remove.Length > 0)
            {
                foreach (var id in c.remove) index.Remove(id);                    // HNSWIndex.cs:83-92, in the given order
                var after = new List<KNNResult<float[], float>>[c.nq];
                for (int i = 0; i < c.nq; i++) after[i] = index.KnnQuery(q[i], c.k);
                WriteResults(w, "knn_after_remove", after);
                w.WriteNumber("count_after_remove", index.Count);
                w.WriteStartArray("ids_after_remove"); foreach (var id in index.Ids()) w.WriteNumberValue(id); w.WriteEndArray();   // :242
                index.Serialize(Path.Combine(outDir, c.name + ".after_remove.snapshot"));
            }
            w.WriteEndObject();
            Console.WriteLine($"{c.name}: {c.n} x {c.dim} {c.metric}, {c.nq} queries");
        }
        return 0;
    }
}
