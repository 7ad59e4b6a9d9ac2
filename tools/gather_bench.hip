// gather_bench.hip — what random dependent 16-byte gathers cost on this chip (the access pattern of an
// FM-index step): every lane chases its own pseudo-random chain through a table of 64-byte records.
//   ./gather_bench <table_MB> <mode> <steps> [lanes] [alloc]
//   alloc 0 = hipMalloc (default); 1 = one physical allocation mapped through the virtual-memory API (hipMemCreate / hipMemMap,
//           address reserved with 1 GB alignment): does the table's page size / TLB reach change the gather rate?
//   mode 1|2|3 : that many loads inside ONE random 64-byte record per step (16 B; 8 B + 16 B; 16 + 16 + 16 B)
//   mode 4     : two 16-byte loads per step, one in each 64-byte half of ONE random 128-byte aligned line
//   mode 5     : two 16-byte loads per step in two INDEPENDENT random 64-byte records
//   mode 6     : one 4-byte load per step (random dword)
// Reports G steps/s and GB/s of touched 64-byte records.  Used to set the practical roofline in DESIGN.md; modes 1, 4, 5
// under `rocprofv3 --pmc TCC_EA0_RDREQ_sum FETCH_SIZE` calibrate how many bytes a fabric read request of a random gather
// carries (tools/gather_calib.sh): the number of records touched is known exactly (lanes x steps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int LOADS>
__global__ void chase(const uint64_t *tab, uint64_t nrec, int steps, uint64_t *out) {
  uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  x = x * 0x9E3779B97F4A7C15ull + 12345;
  uint64_t acc = 0;
  for (int s = 0; s < steps; ++s) {
    const uint64_t r = (x >> 11) % nrec;
    const uint64_t *rec = tab + r * 8;
    uint64_t v;
    if (LOADS == 1) {
      const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(rec + 2 * (x & 3));
      v = a.x ^ a.y;
    } else if (LOADS == 2) {
      const uint64_t m = rec[x & 3];
      const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(rec + 4 + 2 * ((x >> 2) & 1));
      v = m ^ a.x ^ a.y;
    } else if (LOADS == 3) {
      const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(rec);
      const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(rec + 2);
      const ulonglong2 c = *reinterpret_cast<const ulonglong2 *>(rec + 4 + 2 * ((x >> 2) & 1));
      v = a.x ^ a.y ^ b.x ^ b.y ^ c.x ^ c.y;
    } else if (LOADS == 4) {      // both halves of one 128-byte line
      const uint64_t *line = tab + (r & ~1ull) * 8;
      const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(line + 2 * (x & 3));
      const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(line + 8 + 2 * ((x >> 2) & 3));
      v = a.x ^ a.y ^ b.x ^ b.y;
    } else if (LOADS == 5) {      // two independent records
      const uint64_t r2 = ((x * 0xD1342543DE82EF95ull) >> 11) % nrec;
      const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(rec + 2 * (x & 3));
      const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(tab + r2 * 8 + 2 * ((x >> 2) & 3));
      v = a.x ^ a.y ^ b.x ^ b.y;
    } else {                      // one dword
      v = reinterpret_cast<const uint32_t *>(rec)[x & 15];
    }
    acc += v;
    x = x * 6364136223846793005ull + 1442695040888963407ull + v;   // next address depends on the loaded data
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void fill(uint64_t *tab, uint64_t nwords) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * blockDim.x)
    tab[i] = i * 0x9E3779B97F4A7C15ull;
}

int main(int argc, char **argv) {
  const uint64_t mb = argc > 1 ? strtoull(argv[1], 0, 10) : 512;
  const int loads = argc > 2 ? atoi(argv[2]) : 2;
  const int steps = argc > 3 ? atoi(argv[3]) : 100;
  const uint64_t lanes = argc > 4 ? strtoull(argv[4], 0, 10) : (uint64_t)256 * 2048 * 8;
  const uint64_t nrec = mb * 1024 * 1024 / 64;
  const int alloc = argc > 5 ? atoi(argv[5]) : 0;
  uint64_t *tab, *out;
  if (alloc == 1) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t bytes = ((nrec * 64 + gran - 1) / gran) * gran;
    hipMemGenericAllocationHandle_t h;
    CHECK(hipMemCreate(&h, bytes, &prop, 0));
    void *va = nullptr;
    CHECK(hipMemAddressReserve(&va, bytes, (size_t)1 << 30, nullptr, 0));
    CHECK(hipMemMap(va, bytes, 0, h, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CHECK(hipMemSetAccess(va, bytes, &acc, 1));
    tab = (uint64_t *)va;
    printf("vmm: granularity %zu bytes, va %p\n", gran, va);
  } else
  CHECK(hipMalloc(&tab, nrec * 64));
  CHECK(hipMalloc(&out, lanes * 8));
  fill<<<4096, 256>>>(tab, nrec * 8);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    const unsigned grid = (unsigned)(lanes / 256);
    if (loads == 1) chase<1><<<grid, 256>>>(tab, nrec, steps, out);
    else if (loads == 2) chase<2><<<grid, 256>>>(tab, nrec, steps, out);
    else if (loads == 3) chase<3><<<grid, 256>>>(tab, nrec, steps, out);
    else if (loads == 4) chase<4><<<grid, 256>>>(tab, nrec, steps, out);
    else if (loads == 5) chase<5><<<grid, 256>>>(tab, nrec, steps, out);
    else chase<6><<<grid, 256>>>(tab, nrec, steps, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double st = (double)lanes * steps;
    if (rep == 2)
      printf("table %6lu MB  loads/step %d  lanes %lu  steps %d : %8.3f ms  %7.2f G steps/s  %8.1f GB/s of 64B records\n",
             (unsigned long)mb, loads, (unsigned long)lanes, steps, ms, st / ms / 1e6, st * 64 / ms / 1e6);
  }
  return 0;
}
