// gather.hip — what a CU can do with DIVERGENT record fetches (the access pattern of BVH traversal):
// every lane follows its own pointer chain through a table of records, `REQ` x 16 bytes per record,
// `CHAINS` independent chains per lane, W waves per CU.  Reports lane-fetches per CU-clock and the
// time of one dependent step.  Not part of the product; built by `make bin/gather_bench`.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

// record r: word 0 of its first uint4 = index of the next record of the chain (a random permutation cycle)
template <int REQ, int CHAINS>
__global__ __launch_bounds__(1024) void k_chase(const uint4* __restrict__ table, uint32_t recQuads, uint32_t nrec, uint32_t steps,
                                               uint32_t coherent, uint32_t* __restrict__ sink) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t cur[CHAINS];
  uint32_t acc = 0;
  for (int c = 0; c < CHAINS; ++c) {
    // coherent = 1: the lanes of a wave start on the same record (every fetch is a broadcast)
    const uint32_t seed = coherent ? (gid >> 6) : gid;
    cur[c] = (uint32_t)(((uint64_t)(seed * 2654435761u + c * 40503u) * 2246822519u) % nrec);
  }
  for (uint32_t s = 0; s < steps; ++s) {
    uint4 v[CHAINS][REQ];
    for (int c = 0; c < CHAINS; ++c)
      for (int q = 0; q < REQ; ++q) v[c][q] = table[(size_t)cur[c] * recQuads + q];
    for (int c = 0; c < CHAINS; ++c) {
      cur[c] = v[c][0].x;
      for (int q = 0; q < REQ; ++q) acc += v[c][q].y ^ v[c][q].w;
    }
  }
  if (acc == 0x12345678u) sink[gid] = acc;  // (keeps the loads alive)
  if (gid == 0) sink[0] = cur[0];
}

template <int REQ, int CHAINS>
static float run(const uint4* table, uint32_t recQuads, uint32_t nrec, uint32_t steps, uint32_t coherent, uint32_t* sink, int cus, int waves) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_chase<REQ, CHAINS>), dim3(cus), dim3(64 * waves), 0, nullptr, table, recQuads, nrec, 8u, coherent, sink);  // warm
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k_chase<REQ, CHAINS>), dim3(cus), dim3(64 * waves), 0, nullptr, table, recQuads, nrec, steps, coherent, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms;
}


// Cooperative fetch: the G = REQ lanes of a group fetch each other's records together — instruction j loads the record
// of the group's lane j, every lane one 16-B quad of it (adjacent lanes, adjacent quads of one record).  The same bytes
// per lane as k_chase<REQ, 1>, but every instruction touches 64 / G cache lines instead of 64: does the vector L1 spend
// its tag lookups per LANE or per distinct LINE of neighbouring lanes?
template <int G>
__global__ __launch_bounds__(1024) void k_chase_coop(const uint4* __restrict__ table, uint32_t nrec, uint32_t steps, uint32_t* __restrict__ sink) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u, sub = lane % G, base = lane - sub;
  uint32_t cur = (uint32_t)(((uint64_t)(gid * 2654435761u) * 2246822519u) % nrec);
  uint32_t acc = 0;
  for (uint32_t s = 0; s < steps; ++s) {
    uint4 v[G];
    for (int j = 0; j < G; ++j) {
      const uint32_t rec = (uint32_t)__shfl((int)cur, (int)(base + j));
      v[j] = table[(size_t)rec * G + sub];
    }
    uint32_t next = 0;
    for (int j = 0; j < G; ++j) {
      const uint32_t w = (uint32_t)__shfl((int)v[j].x, (int)base);  // word 0 of quad 0 of lane j's record
      next = sub == (uint32_t)j ? w : next;
      acc += v[j].y ^ v[j].w;
    }
    cur = next;
  }
  if (acc == 0x12345678u) sink[gid] = acc;
  if (gid == 0) sink[0] = cur;
}

template <int G>
static float run_coop(const uint4* table, uint32_t nrec, uint32_t steps, uint32_t* sink, int cus, int waves) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_chase_coop<G>), dim3(cus), dim3(64 * waves), 0, nullptr, table, nrec, 8u, sink);  // warm
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k_chase_coop<G>), dim3(cus), dim3(64 * waves), 0, nullptr, table, nrec, steps, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

// `gather_bench one <table KB> <record bytes>`: one configuration (2 requests per record, one chain, 16 waves per CU) and
// nothing else — what a rocprofv3 --pmc FETCH_SIZE pass is pointed at to see the HBM-side bytes a random record costs.
static int one(size_t tableKB, uint32_t recBytes);

int main(int argc, char** argv) {
  if (argc >= 4 && !strcmp(argv[1], "one")) return one((size_t)atoll(argv[2]), (uint32_t)atoi(argv[3]));
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  uint32_t* sink;
  CK(hipMalloc(&sink, 256 * 1024 * 4 * 4));
  const double ghz = 2.4;
  printf("{\"cus\": %d, \"rows\": [\n", cus);
  bool first = true;
  for (size_t tableKB : {8u, 24u, 256u, 1024u, 16384u, 131072u, 1048576u}) {
    const size_t tableMB = tableKB / 1024u;
    for (uint32_t recBytes : {32u, 64u}) {
      const uint32_t recQuads = recBytes / 16u;
      const uint32_t nrec = (uint32_t)(tableKB * 1024u / recBytes);
      std::vector<uint4> h((size_t)nrec * recQuads);
      // one random cycle through all records (Sattolo)
      std::vector<uint32_t> perm(nrec);
      for (uint32_t i = 0; i < nrec; ++i) perm[i] = i;
      uint64_t st = 88172645463325252ull;
      auto rnd = [&]() { st ^= st << 13, st ^= st >> 7, st ^= st << 17; return st; };
      for (uint32_t i = nrec - 1; i > 0; --i) std::swap(perm[i], perm[rnd() % i]);
      for (uint32_t i = 0; i < nrec; ++i)
        for (uint32_t q = 0; q < recQuads; ++q) h[(size_t)i * recQuads + q] = make_uint4(perm[i], i, q, 7u);
      uint4* d;
      CK(hipMalloc(&d, h.size() * sizeof(uint4)));
      CK(hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
      for (int waves : {8, 16}) {
        for (int variant = 0; variant < 6; ++variant) {
          // variant: (requests per record, chains per lane, coherent)
          static const int REQS[6] = {1, 2, 4, 2, 2, 2}, CH[6] = {1, 1, 1, 2, 4, 1}, COH[6] = {0, 0, 0, 0, 0, 1};
          const int req = REQS[variant], ch = CH[variant], coh = COH[variant];
          if ((uint32_t)req > recQuads) continue;
          const uint32_t steps = tableMB >= 128 ? 512u : 2048u;
          float ms = 0;
          if (req == 1 && ch == 1) ms = run<1, 1>(d, recQuads, nrec, steps, coh, sink, cus, waves);
          else if (req == 2 && ch == 1) ms = run<2, 1>(d, recQuads, nrec, steps, coh, sink, cus, waves);
          else if (req == 4 && ch == 1) ms = run<4, 1>(d, recQuads, nrec, steps, coh, sink, cus, waves);
          else if (req == 2 && ch == 2) ms = run<2, 2>(d, recQuads, nrec, steps, coh, sink, cus, waves);
          else if (req == 2 && ch == 4) ms = run<2, 4>(d, recQuads, nrec, steps, coh, sink, cus, waves);
          const double cyc = ms * 1e-3 * ghz * 1e9;
          const double stepCycles = cyc / steps;                                  // one dependent step of a wave
          const double laneReqPerClk = (double)waves * 64 * ch * req * steps / cyc;  // per CU
          const double recPerSec = (double)cus * waves * 64 * ch * steps / (ms * 1e-3);
          printf("%s{\"table_kb\": %zu, \"rec_bytes\": %u, \"waves_per_cu\": %d, \"req_per_rec\": %d, \"chains\": %d, \"coherent\": %d, "
                 "\"ms\": %.3f, \"step_cycles\": %.0f, \"lane_req_per_clk_per_cu\": %.3f, \"grec_per_s\": %.1f, \"gbs\": %.0f}",
                 first ? "" : ",\n", tableKB, recBytes, waves, req, ch, coh, ms, stepCycles, laneReqPerClk, recPerSec / 1e9,
                 recPerSec * req * 16 / 1e9);
          first = false;
        }
        {
          // the cooperative fetch of the same records (REQ = lanes per group = quads per record)
          const uint32_t steps = tableMB >= 128 ? 512u : 2048u;
          const float ms = recQuads == 2 ? run_coop<2>(d, nrec, steps, sink, cus, waves) : run_coop<4>(d, nrec, steps, sink, cus, waves);
          const double cyc = ms * 1e-3 * ghz * 1e9;
          const double recPerSec = (double)cus * waves * 64 * steps / (ms * 1e-3);
          printf(",\n{\"table_kb\": %zu, \"rec_bytes\": %u, \"waves_per_cu\": %d, \"req_per_rec\": %u, \"chains\": 1, \"coherent\": 0, \"cooperative_lanes\": %u, "
                 "\"ms\": %.3f, \"step_cycles\": %.0f, \"lane_req_per_clk_per_cu\": %.3f, \"grec_per_s\": %.1f, \"gbs\": %.0f}",
                 tableKB, recBytes, waves, recQuads, recQuads, ms, cyc / steps, (double)waves * 64 * recQuads * steps / cyc, recPerSec / 1e9,
                 recPerSec * recBytes / 1e9);
        }
      }
      CK(hipFree(d));
    }
  }
  printf("\n]}\n");
  return 0;
}

static int one(size_t tableKB, uint32_t recBytes) {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  uint32_t* sink;
  CK(hipMalloc(&sink, 256 * 1024 * 4 * 4));
  const uint32_t recQuads = recBytes / 16u, nrec = (uint32_t)(tableKB * 1024u / recBytes);
  std::vector<uint4> h((size_t)nrec * recQuads);
  std::vector<uint32_t> perm(nrec);
  for (uint32_t i = 0; i < nrec; ++i) perm[i] = i;
  uint64_t st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13, st ^= st >> 7, st ^= st << 17; return st; };
  for (uint32_t i = nrec - 1; i > 0; --i) std::swap(perm[i], perm[rnd() % i]);
  for (uint32_t i = 0; i < nrec; ++i)
    for (uint32_t q = 0; q < recQuads; ++q) h[(size_t)i * recQuads + q] = make_uint4(perm[i], i, q, 7u);
  uint4* d;
  CK(hipMalloc(&d, h.size() * sizeof(uint4)));
  CK(hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
  const uint32_t steps = 1024;
  const float ms = run<2, 1>(d, recQuads, nrec, steps, 0, sink, cus, 16);
  const double recs = (double)cus * 16 * 64 * steps;
  printf("{\"table_kb\": %zu, \"rec_bytes\": %u, \"steps\": %u, \"records\": %.0f, \"ms\": %.3f, \"grec_per_s\": %.1f}\n", tableKB, recBytes, steps, recs,
         ms, recs / (ms * 1e-3) / 1e9);
  return 0;
}
