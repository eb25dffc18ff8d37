// tools/gather_bench.hip -- measured ceiling of RANDOM ROW GATHERS on this chip, per row size (development tool).
//
// The traversal kernels read one candidate row per distance evaluation at a random place of a table far larger
// than the 256-MiB Infinity Cache.  What such a pattern can reach at best is not the 8 TB/s HBM peak (nor the
// 6.3 TB/s of a streamed copy): this tool measures it for the three row sizes of BASELINE.json's configurations
// -- 128-B int8 records (C5), 512-B rows (C2 / C4), 3 072-B rows (C3) -- with the product's lane mapping (8 lanes
// per row, strided dwords, up to 64 loads in flight per lane) and, for the records, with one 16-byte load per lane.
// bench.py quotes `frac_of_measured_gather` against these numbers; the same binary under
// `rocprofv3 --pmc FETCH_SIZE` calibrates the counter on a known byte count in this access pattern.
//
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o tools/gather_bench
//   tools/gather_bench <row_bytes: 128|512|3072> [table_GiB = 4] [rows_per_launch = 16M] [variant = 0]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e = (x);                                                                    \
        if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); }        \
    } while (0)

// 8 lanes per row, lane j reads words j, j + 8, ... ; a wave has 8 rows in flight per pass and PASSES passes issued
// before anything is consumed (WORDS / 8 * PASSES loads in flight per lane, capped at 64 as in the product)
template <int WORDS, int PASSES>
__global__ void __launch_bounds__(256) gather_dwords(const float *__restrict__ table, const int *__restrict__ ids, float *__restrict__ out, long long nrows)
{
    const int lane = threadIdx.x & 63, grp = lane >> 3, j = lane & 7;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    float acc = 0.f;
    constexpr int CH = WORDS > 128 ? 128 : WORDS; // words of a row requested at once (the product chunks long rows the same way)
    for (long long r0 = wave * 8 * PASSES; r0 < nrows; r0 += nwaves * 8 * PASSES) {
        int id[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) { const long long r = r0 + p * 8 + grp; id[p] = ids[r < nrows ? r : 0]; }
        for (int c0 = 0; c0 < WORDS; c0 += CH) {
            float v[PASSES][CH / 8];
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const float *row = table + (size_t)id[p] * WORDS + c0;
#pragma unroll
                for (int k = 0; k < CH / 8; ++k) v[p][k] = row[8 * k + j];
            }
#pragma unroll
            for (int p = 0; p < PASSES; ++p)
#pragma unroll
                for (int k = 0; k < CH / 8; ++k) acc += v[p][k];
        }
    }
    if (acc == 123.456f) out[0] = acc; // keeps the loads
}

// 128-B records: 8 lanes per record, ONE 16-byte load per lane; 8 * PASSES records in flight per wave
template <int PASSES>
__global__ void __launch_bounds__(256) gather_rec16(const float4 *__restrict__ table, const int *__restrict__ ids, float *__restrict__ out, long long nrows)
{
    const int lane = threadIdx.x & 63, grp = lane >> 3, j = lane & 7;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    float acc = 0.f;
    for (long long r0 = wave * 8 * PASSES; r0 < nrows; r0 += nwaves * 8 * PASSES) {
        float4 v[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) { const long long r = r0 + p * 8 + grp; v[p] = table[(size_t)ids[r < nrows ? r : 0] * 8 + j]; }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) acc += v[p].x + v[p].y + v[p].z + v[p].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

// 512-B rows with 16-byte loads.  LPR = 2: two lanes per row, lane h reads the float4 at words 8k + 4h (the AVX lanes
// 4h..4h+3 of block k: the reference's summation order survives), 16 loads per lane, 32 rows per pass.  LPR = 8: eight
// lanes per row, lane j reads the float4 at words 32i + 4j (whole 128-B lines per instruction; the summation order would
// need a chain across lanes), 4 loads per lane and pass, PASSES passes in flight.
template <int LPR, int PASSES>
__global__ void __launch_bounds__(256) gather_vec4_512(const float4 *__restrict__ table, const int *__restrict__ ids, float *__restrict__ out, long long nrows)
{
    constexpr int RPP = 64 / LPR, LOADS = 32 / LPR; // rows per pass, float4 loads per lane and row
    const int lane = threadIdx.x & 63, grp = lane / LPR, j = lane % LPR;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    float acc = 0.f;
    for (long long r0 = wave * RPP * PASSES; r0 < nrows; r0 += nwaves * RPP * PASSES) {
        float4 v[PASSES][LOADS];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const long long r = r0 + p * RPP + grp;
            const float4 *row = table + (size_t)ids[r < nrows ? r : 0] * 32;
#pragma unroll
            for (int k = 0; k < LOADS; ++k) v[p][k] = row[LPR * k + j];
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p)
#pragma unroll
            for (int k = 0; k < LOADS; ++k) acc += v[p][k].x + v[p][k].y + v[p][k].z + v[p][k].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char **argv)
{
    const int row_bytes = argc > 1 ? atoi(argv[1]) : 512;
    const double gib = argc > 2 ? atof(argv[2]) : 4.0;
    const long long nrows = argc > 3 ? atoll(argv[3]) : (16LL << 20);
    const int variant = argc > 4 ? atoi(argv[4]) : 0;
    if (row_bytes != 128 && row_bytes != 512 && row_bytes != 3072) { printf("row_bytes must be 128, 512 or 3072\n"); return 1; }
    const long long table_rows = (long long)(gib * 1073741824.0 / row_bytes);
    float *table, *out;
    int *ids;
    CK(hipMalloc(&table, (size_t)table_rows * row_bytes));
    CK(hipMemset(table, 0, (size_t)table_rows * row_bytes));
    CK(hipMalloc(&out, 64));
    std::vector<int> h((size_t)nrows);
    unsigned long long s = 88172645463325252ull;
    for (long long i = 0; i < nrows; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[(size_t)i] = (int)(s % (unsigned long long)table_rows); }
    CK(hipMalloc(&ids, sizeof(int) * (size_t)nrows));
    CK(hipMemcpy(ids, h.data(), sizeof(int) * (size_t)nrows, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    int dev = 0, cus = 256;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const char *names[5] = {"8 lanes x strided dwords (the product's mapping)", "8 lanes x one 16-byte load (records only)", "2 lanes x 16 float4 loads, 32 rows per pass", "8 lanes x 4 float4 loads x 4 passes (whole lines)", "2 lanes x 16 float4 loads x 2 passes"};
    for (int grid_mult : {4, 8, 16}) {
        const int grid = cus * grid_mult;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(a));
            if (row_bytes == 512 && variant == 2) hipLaunchKernelGGL((gather_vec4_512<2, 1>), dim3(grid), dim3(256), 0, 0, (const float4 *)table, ids, out, nrows);
            else if (row_bytes == 512 && variant == 3) hipLaunchKernelGGL((gather_vec4_512<8, 4>), dim3(grid), dim3(256), 0, 0, (const float4 *)table, ids, out, nrows);
            else if (row_bytes == 512 && variant == 4) hipLaunchKernelGGL((gather_vec4_512<2, 2>), dim3(grid), dim3(256), 0, 0, (const float4 *)table, ids, out, nrows);
            else if (row_bytes == 128 && variant == 1) hipLaunchKernelGGL(gather_rec16<8>, dim3(grid), dim3(256), 0, 0, (const float4 *)table, ids, out, nrows);
            else if (row_bytes == 128) hipLaunchKernelGGL((gather_dwords<32, 8>), dim3(grid), dim3(256), 0, 0, table, ids, out, nrows);
            else if (row_bytes == 512) hipLaunchKernelGGL((gather_dwords<128, 4>), dim3(grid), dim3(256), 0, 0, table, ids, out, nrows);
            else hipLaunchKernelGGL((gather_dwords<768, 4>), dim3(grid), dim3(256), 0, 0, table, ids, out, nrows);
            CK(hipGetLastError());
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("gather row_bytes=%d table=%.1fGiB rows=%lld variant=%d [%s] blocks_per_cu=%d: %.3f ms  %.1f GB/s  %.2f G rows/s  (bytes per launch %lld)\n",
               row_bytes, gib, nrows, variant, names[variant], grid_mult, best, (double)nrows * row_bytes / best / 1e6, (double)nrows / best / 1e6,
               nrows * (long long)row_bytes);
    }
    return 0;
}
