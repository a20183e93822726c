// Boundary-cost probe: the same dependent kernel chains submitted (i) as a hipGraph replay and (ii) as raw AQL
// kernel-dispatch packets on a user-mode HSA queue with the barrier bit set and a chosen acquire / release fence scope per packet.
// Question: is a dependent kernel boundary cheaper than the runtime's packet form (1.45-1.9 us) when the packet carries narrower
// (or no) cache fences and the kernels keep their hand-offs coherent themselves (sc1 loads / write-through stores)?
//
// build (tools/ubench/aql/build.sh): hipcc -O2 aql_probe.cpp -o aql_probe -lhsa-runtime64 ; device code: aql_kernels.hsaco
// run: ./aql_probe aql_kernels.hsaco
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define HCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP %s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define SCK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); printf("HSA %s: %s\n", #x, m_ ? m_ : "?"); exit(1); } } while (0)

static hsa_agent_t g_gpu, g_cpu;
static bool g_have_gpu = false, g_have_cpu = false;
static hsa_amd_memory_pool_t g_kernarg_pool, g_dev_pool;
static bool g_have_kpool = false, g_have_dpool = false;

static hsa_status_t agent_cb(hsa_agent_t a, void*) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t cpu_pool_cb(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t flags = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_kpool) { g_kernarg_pool = p; g_have_kpool = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t gpu_pool_cb(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t flags = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_dpool) { g_dev_pool = p; g_have_dpool = true; }
  return HSA_STATUS_SUCCESS;
}

struct Kern {
  uint64_t object = 0;
  uint32_t kernarg = 0, group = 0, priv = 0;
  hipFunction_t hip = nullptr;
};

struct Launch {   // one dispatch of a chain
  const Kern* k;
  uint32_t grid;          // workgroups of 256 threads
  unsigned char args[64];
  uint32_t nargs;         // bytes
};

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct AqlQueue {
  hsa_queue_t* q = nullptr;
  hsa_signal_t done{};
  char* kernarg_host = nullptr;   // fine-grained host kernarg pool
  char* kernarg_dev = nullptr;    // device memory (filled through a staging copy)
  size_t kernarg_bytes = 0;
};

// submit `reps` x chain, one doorbell; returns us per launch (host wall from doorbell to completion of the last packet)
static double run_aql(AqlQueue& Q, const std::vector<Launch>& chain, int reps, int acq, int rel, int barrier, bool dev_kernarg) {
  const size_t n = chain.size() * (size_t)reps;
  if (n > Q.q->size) { printf("queue too small\n"); exit(1); }
  const size_t stride = 64;
  if (n * stride > Q.kernarg_bytes) { printf("kernarg pool too small\n"); exit(1); }
  std::vector<char> stage(n * stride, 0);
  for (size_t i = 0; i < n; ++i) memcpy(stage.data() + i * stride, chain[i % chain.size()].args, chain[i % chain.size()].nargs);
  char* kbase = dev_kernarg ? Q.kernarg_dev : Q.kernarg_host;
  if (dev_kernarg) HCK(hipMemcpy(Q.kernarg_dev, stage.data(), stage.size(), hipMemcpyHostToDevice));
  else memcpy(Q.kernarg_host, stage.data(), stage.size());
  hsa_signal_store_relaxed(Q.done, 1);
  const uint64_t base = hsa_queue_add_write_index_relaxed(Q.q, n);
  while (base + n - hsa_queue_load_read_index_scacquire(Q.q) > Q.q->size) {}
  const uint32_t mask = Q.q->size - 1;
  hsa_kernel_dispatch_packet_t* ring = (hsa_kernel_dispatch_packet_t*)Q.q->base_address;
  for (size_t i = 0; i < n; ++i) {
    const Launch& L = chain[i % chain.size()];
    hsa_kernel_dispatch_packet_t* p = ring + ((base + i) & mask);
    p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
    p->reserved0 = 0;
    p->grid_size_x = L.grid * 256; p->grid_size_y = 1; p->grid_size_z = 1;
    p->private_segment_size = L.k->priv;
    p->group_segment_size = L.k->group;
    p->kernel_object = L.k->object;
    p->kernarg_address = kbase + i * stride;
    p->reserved2 = 0;
    p->completion_signal.handle = (i + 1 == n) ? Q.done.handle : 0;
    // first packet of the batch acquires at system scope (the kernargs / buffers were written by the host), the last releases
    // at system scope; everything in between carries the scopes under test
    const int a = (i == 0) ? HSA_FENCE_SCOPE_SYSTEM : acq;
    const int r = (i + 1 == n) ? HSA_FENCE_SCOPE_SYSTEM : rel;
    const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                            (a << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (r << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
  }
  const double t0 = now_us();
  hsa_signal_store_screlease(Q.q->doorbell_signal, base + n - 1);
  while (hsa_signal_wait_scacquire(Q.done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) {}
  const double t1 = now_us();
  return (t1 - t0) / (double)n;
}

static double run_graph(hipStream_t st, const std::vector<Launch>& chain, int reps) {
  hipGraph_t g; hipGraphExec_t ge;
  HCK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (const Launch& L : chain) {
    size_t sz = L.nargs;
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, (void*)L.args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    HCK(hipModuleLaunchKernel(L.k->hip, L.grid, 1, 1, 256, 1, 1, 0, st, nullptr, cfg));
  }
  HCK(hipStreamEndCapture(st, &g));
  HCK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  HCK(hipGraphLaunch(ge, st)); HCK(hipStreamSynchronize(st));
  hipEvent_t e0, e1; HCK(hipEventCreate(&e0)); HCK(hipEventCreate(&e1));
  HCK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) HCK(hipGraphLaunch(ge, st));
  HCK(hipEventRecord(e1, st)); HCK(hipStreamSynchronize(st));
  float ms; HCK(hipEventElapsedTime(&ms, e0, e1));
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1000.0 / ((double)reps * chain.size());
}

template <typename... T>
static Launch mk(const Kern& k, uint32_t grid, T... a) {
  Launch L{};
  L.k = &k; L.grid = grid; L.nargs = 0;
  auto put = [&](const void* p, size_t n, size_t al) { L.nargs = (uint32_t)((L.nargs + al - 1) / al * al); memcpy(L.args + L.nargs, p, n); L.nargs += (uint32_t)n; };
  (void)put;
  int dummy[] = {0, (put(&a, sizeof(a), sizeof(a)), 0)...};
  (void)dummy;
  return L;
}

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "aql_kernels.hsaco";
  HCK(hipSetDevice(0));
  hipStream_t st; HCK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  SCK(hsa_init());
  SCK(hsa_iterate_agents(agent_cb, nullptr));
  if (!g_have_gpu || !g_have_cpu) { printf("no agents\n"); return 1; }
  SCK(hsa_amd_agent_iterate_memory_pools(g_cpu, cpu_pool_cb, nullptr));
  SCK(hsa_amd_agent_iterate_memory_pools(g_gpu, gpu_pool_cb, nullptr));
  if (!g_have_kpool) { printf("no kernarg pool\n"); return 1; }
  char name[64] = "";
  hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, name);
  printf("# agent %s\n", name);

  // ---- code object: HSA loader + HIP module of the same file
  FILE* f = fopen(path, "rb");
  if (!f) { printf("cannot open %s\n", path); return 1; }
  fseek(f, 0, SEEK_END); const long fsz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<char> blob(fsz);
  if (fread(blob.data(), 1, fsz, f) != (size_t)fsz) return 1;
  fclose(f);
  hsa_code_object_reader_t reader; hsa_executable_t exe;
  SCK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader));
  SCK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  SCK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
  SCK(hsa_executable_freeze(exe, ""));
  hipModule_t mod; HCK(hipModuleLoadData(&mod, blob.data()));
  auto get = [&](const char* nm) {
    Kern k;
    hsa_executable_symbol_t sym;
    SCK(hsa_executable_get_symbol_by_name(exe, (std::string(nm) + ".kd").c_str(), &g_gpu, &sym));
    SCK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    SCK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
    SCK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    SCK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    HCK(hipModuleGetFunction(&k.hip, mod, nm));
    return k;
  };
  const Kern k_empty = get("k_empty"), k_inc = get("k_inc"), k_inc_sc1 = get("k_inc_sc1"), k_all = get("k_all"), k_all_sc1 = get("k_all_sc1"),
             k_stream = get("k_stream");

  AqlQueue Q;
  SCK(hsa_queue_create(g_gpu, 16384, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &Q.q));
  SCK(hsa_signal_create(1, 0, nullptr, &Q.done));
  Q.kernarg_bytes = 16384 * 64;
  SCK(hsa_amd_memory_pool_allocate(g_kernarg_pool, Q.kernarg_bytes, 0, (void**)&Q.kernarg_host));
  SCK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, Q.kernarg_host));
  HCK(hipMalloc(&Q.kernarg_dev, Q.kernarg_bytes));

  const int G = 1024;
  float *a, *b;
  HCK(hipMalloc(&a, G * 256 * 4)); HCK(hipMalloc(&b, G * 256 * 4));
  std::vector<float> host(G * 256);
  const int scopes[3] = {HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_SYSTEM};
  const char* sn[3] = {"none", "agent", "system"};

  // ---- 1. trivial chains: cost per dependent launch by packet form; correctness of the hand-off checked on the final values
  const int N = 600, R = 20;
  for (int grid : {256, 1024}) {
    for (int kind = 0; kind < 5; ++kind) {
      const Kern& k = kind == 0 ? k_empty : kind == 1 ? k_inc : kind == 2 ? k_inc_sc1 : kind == 3 ? k_all : k_all_sc1;
      const char* kn = kind == 0 ? "empty" : kind == 1 ? "inc" : kind == 2 ? "inc_sc1" : kind == 3 ? "all" : "all_sc1";
      std::vector<Launch> chain;
      for (int i = 0; i < N; ++i) {
        const float* in = (i & 1) ? b : a; float* out = (i & 1) ? a : b;
        chain.push_back(kind >= 3 ? mk(k, grid, in, out, grid) : mk(k, grid, in, out));
      }
      // expected value after reps x N launches: inc -> +1 per launch on every element; all -> mean + 1 (all elements equal -> +1)
      auto reset = [&]() { HCK(hipMemset(a, 0, G * 256 * 4)); HCK(hipMemset(b, 0, G * 256 * 4)); HCK(hipDeviceSynchronize()); };
      auto check = [&](int launches) -> int {
        if (kind == 0) return 0;
        HCK(hipDeviceSynchronize());
        HCK(hipMemcpy(host.data(), (launches & 1) ? b : a, G * 256 * 4, hipMemcpyDeviceToHost));
        const int cnt = kind >= 3 ? grid : grid * 256;
        int bad = 0;
        for (int i = 0; i < cnt; ++i) bad += host[i] != (float)launches;
        return bad;
      };
      reset();
      const double tg = run_graph(st, chain, R);
      printf("grid %4d %-8s hipGraph            : %.3f us/launch\n", grid, kn, tg);
      for (int dk = 0; dk < 2; ++dk)
        for (int ai = 0; ai < 3; ++ai)
          for (int ri = 0; ri < 3; ++ri) {
            if (dk == 1 && !(ai == ri)) continue;   // device kernargs: the diagonal only
            reset();
            run_aql(Q, chain, 2, scopes[ai], scopes[ri], 1, dk);   // warm
            reset();
            const double t = run_aql(Q, chain, R, scopes[ai], scopes[ri], 1, dk);
            const int bad = check(N * R);
            printf("grid %4d %-8s aql acq=%-6s rel=%-6s kernarg=%s : %.3f us/launch   stale/wrong values %d\n", grid, kn, sn[ai], sn[ri], dk ? "dev " : "host", t, bad);
          }
      reset();
      const double t0 = run_aql(Q, chain, R, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 0, 0);
      printf("grid %4d %-8s aql NO barrier bit (independent dispatch rate) : %.3f us/launch\n", grid, kn, t0);
      fflush(stdout);
    }
  }

  // ---- 2. a decoder layer-pass stand-in: four dependent weight-streaming launches (3 / 2 / 33.5 / 16.8 MB on 192 / 128 / 1024 / 512
  // workgroups), every launch reading the previous one's whole output vector; weights cycle through a 1.5 GiB pool
  {
    const size_t pool = (size_t)1536 << 20;
    char* W; HCK(hipMalloc(&W, pool)); HCK(hipMemset(W, 1, pool));
    const size_t mbs[4] = {3u << 20, 2u << 20, (size_t)(33.5 * (1 << 20)), (size_t)(16.8 * (1 << 20))};
    const int grids[4] = {192, 128, 1024, 512};
    for (int sc1 = 0; sc1 < 2; ++sc1) {
      std::vector<Launch> chain;
      size_t off = 0;
      int prev_n = 512;
      for (int i = 0; i < 4 * 150; ++i) {
        const int j = i & 3;
        const unsigned per16 = (unsigned)(mbs[j] / 16 / grids[j]);
        const size_t bytes = (size_t)per16 * 16 * grids[j];
        if (off + bytes > pool) off = 0;
        const float* in = (i & 1) ? b : a; float* out = (i & 1) ? a : b;
        chain.push_back(mk(k_stream, grids[j], (const void*)(W + off), in, out, per16, prev_n, sc1));
        off += (bytes + 4095) & ~(size_t)4095;
        prev_n = grids[j];
      }
      HCK(hipMemset(a, 0, G * 256 * 4)); HCK(hipMemset(b, 0, G * 256 * 4)); HCK(hipDeviceSynchronize());
      const double tg = run_graph(st, chain, 10);
      printf("layer-pass stand-in (sc1 hand-off %d) hipGraph : %.3f us/launch = %.2f us per 4-launch pass\n", sc1, tg, 4 * tg);
      for (int ai = 0; ai < 3; ++ai)
        for (int ri = 0; ri < 3; ++ri) {
          if (!sc1 && (ai == 0 || ri == 0)) continue;   // plain hand-offs need the packet fences
          run_aql(Q, chain, 2, scopes[ai], scopes[ri], 1, 1);
          const double t = run_aql(Q, chain, 10, scopes[ai], scopes[ri], 1, 1);
          printf("layer-pass stand-in (sc1 hand-off %d) aql acq=%-6s rel=%-6s : %.3f us/launch = %.2f us per 4-launch pass\n", sc1, sn[ai], sn[ri], t, 4 * t);
        }
      fflush(stdout);
    }
    HCK(hipFree(W));
  }
  hsa_queue_destroy(Q.q);
  return 0;
}
