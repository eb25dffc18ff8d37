/* tools/host_sanitize/oracle_main.c -- the CPU restatement (oracle/hnsw_oracle.c, test infrastructure) under
 * AddressSanitizer + UBSan: a small index through sequential Add, the batched schedule, the tick schedule, KnnQuery on one and several
 * threads, RangeQuery, removals with slot reuse.  Built with the oracle's own source by tests/test_host_sanitizers.py:
 *   gcc -std=gnu11 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off -mavx2 -mfma \
 *       -Dorc_main_included oracle_main.c -lm -lpthread */
#include "../../oracle/hnsw_oracle.c"

#include <stdio.h>

int main(void)
{
    enum { N = 700, D = 24, Q = 64, K = 10 };
    static float x[N * D], q[Q * D];
    unsigned s = 12345u;
    for (int i = 0; i < N * D; ++i) { s = s * 1664525u + 1013904223u; x[i] = (float)(s >> 8) / 16777216.0f; }
    for (int i = 0; i < Q * D; ++i) { s = s * 1664525u + 1013904223u; q[i] = (float)(s >> 8) / 16777216.0f; }
    static int ids[N], out_ids[Q * K];
    static float out_d[Q * K];
    for (int metric = 0; metric < 4; ++metric) {
        void *h = orc_create(D, metric, 8, 1.0 / 2.0794415416798357, 5, 40, 64 /* forces resizes */, 31337, 1, 1);
        if (!h) { printf("orc_create failed\n"); return 2; }
        orc_set_remove_max_candidates(h, 40);
        orc_add(h, x, 300, ids);
        orc_add_batched_mt(h, x + 300 * D, 250, ids + 300, 128, 3);
        {
            uint64_t st[5];
            orc_add_ticks(h, x + 550 * D, 150, ids + 550, 32, st); /* collection size 64: the allocations of the call resize the arrays before its first tick */
            if (st[2] > 32) { printf("tick schedule: %llu items in flight\n", (unsigned long long)st[2]); return 3; }
        }
        orc_knn_query(h, q, Q, K, out_ids, out_d, 1);
        orc_knn_query(h, q, Q, K, out_ids, out_d, 4);
        int rm[120];
        for (int i = 0; i < 120; ++i) rm[i] = i * 5;
        orc_remove(h, rm, 120);
        orc_add(h, x, 60, ids); /* slot reuse */
        orc_knn_query(h, q, Q, K, out_ids, out_d, 2);
        {
            static int cnt[Q], rids[Q * 64];
            static float rd[Q * 64];
            orc_range_query(h, q, Q, metric == 0 || metric == 3 ? 2.5f : 0.2f, 64, cnt, rids, rd);
        }
        printf("metric %d: count %d, hash %llu\n", metric, orc_count(h), (unsigned long long)orc_graph_hash(h));
        orc_free(h);
    }
    return 0;
}
