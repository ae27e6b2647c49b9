// Gather-rate microbenchmark for gfx950: what does one wave64 `global_load_dword` cost in the texture-address /
// L1 path as a function of WHICH addresses its 64 lanes read? Everything is L1- or L2-resident (one small region per
// CU), so the time is the address path's, not the memory's. 16 waves per CU (the sweep kernel's occupancy), 8
// independent loads per iteration and lane. Output: cycles per wave-instruction per CU at the clock passed on the
// command line.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_rates gather_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

__global__ void __launch_bounds__(64) k_gather(const unsigned* __restrict__ buf, const unsigned* __restrict__ pat,
                                               unsigned region_dwords, int iters, unsigned* out, int width) {
  const unsigned lane = threadIdx.x;
  // per-lane offsets of the 8 loads (dwords); the pattern advances by `step` dwords per iteration inside the region
  unsigned off[8];
  for (int k = 0; k < 8; ++k) off[k] = pat[k * 64 + lane];
  const unsigned step = pat[8 * 64];
  const unsigned mask = region_dwords - 1;
  const unsigned* base = buf + (size_t)(blockIdx.x % 256) * 0;  // all CUs share the region (L2-resident, L1 per CU)
  unsigned acc = 0, pos = 0;
  for (int i = 0; i < iters; ++i) {
    unsigned v[8];
    if (width == 4) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = base[(off[k] + pos) & mask];
    } else if (width == 5) {   // dword loads at 2-byte granularity: offsets count halfwords
      struct __attribute__((packed, aligned(2))) U { unsigned v; };
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ((const U*)((const unsigned short*)base + ((off[k] + pos) & (2 * mask - 1))))->v;
    } else if (width == 2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ((const unsigned short*)base)[(off[k] + pos) & (2 * mask + 1)];
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ((const unsigned char*)base)[(off[k] + pos) & (4 * mask + 3)];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
    pos += step;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

struct Pattern { std::string name; std::vector<unsigned> off; unsigned step; };

int main(int argc, char** argv) {
  const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
  const int iters = 4000;
  const unsigned region = 1u << 12;  // dwords: 16 KB region (L1-resident per CU)
  std::vector<Pattern> pats;
  auto add = [&](const char* name, auto f, unsigned step) {
    Pattern p; p.name = name; p.off.resize(8 * 64 + 1); p.step = step;
    for (int k = 0; k < 8; ++k) for (int l = 0; l < 64; ++l) p.off[k * 64 + l] = f(k, l);
    p.off[8 * 64] = step;
    pats.push_back(p);
  };
  // dword offsets; a 128-byte line = 32 dwords
  add("all lanes one dword", [](int k, int l) { return (unsigned)(k * 32); }, 32);
  add("64 consecutive dwords (2 lines)", [](int k, int l) { return (unsigned)(k * 64 + l); }, 32);
  add("16-lane groups: consecutive dwords, group = line", [](int k, int l) { return (unsigned)(k * 512 + (l / 16) * 32 + (l % 16)); }, 32);
  add("8-lane groups: 8 consecutive dwords (32 B), group = line", [](int k, int l) { return (unsigned)(k * 512 + (l / 8) * 32 + (l % 8)); }, 32);
  add("quads: 4 consecutive dwords, quad = line", [](int k, int l) { return (unsigned)(k * 512 + (l / 4) * 32 + (l % 4)); }, 32);
  add("pairs: 2 consecutive dwords, pair = line", [](int k, int l) { return (unsigned)(k * 1024 + (l / 2) * 32 + (l % 2)); }, 32);
  add("every lane its own line", [](int k, int l) { return (unsigned)(k * 2048 + l * 32); }, 32);
  add("every lane its own 64-byte half line", [](int k, int l) { return (unsigned)(k * 1024 + l * 16); }, 32);
  add("every lane its own 32-byte sector", [](int k, int l) { return (unsigned)(k * 512 + l * 8); }, 32);
  add("quads: 4 sectors (32 B apart) of one line", [](int k, int l) { return (unsigned)(k * 512 + (l / 4) * 32 + (l % 4) * 8); }, 32);
  add("quads: lanes 0,1 line A, lanes 2,3 line B", [](int k, int l) { return (unsigned)(k * 1024 + (l / 4) * 64 + ((l % 4) / 2) * 32 + (l % 2)); }, 32);
  add("quads: 2 x 2 block of a tile (2 dwords x 2 sectors)", [](int k, int l) { return (unsigned)(k * 512 + (l / 4) * 32 + ((l % 4) / 2) * 8 + (l % 2)); }, 32);
  add("16-lane groups all on one line, interleaved over lanes (lane l -> line l % 4)", [](int k, int l) { return (unsigned)(k * 128 + (l % 4) * 32 + (l / 4)); }, 32);
  add("16-lane groups: 4 lines x 4 dwords, lanes 4i..4i+3 one line", [](int k, int l) { return (unsigned)(k * 512 + (l / 16) * 128 + ((l % 16) / 4) * 32 + (l % 4)); }, 32);
  // quad-level rules: quad q reads line q of the load's 2 KB block at the four dword offsets given
  {
    struct Q { const char* name; int o[4]; };
    static const Q qs[] = {
      {"quad (0,1,2,3)", {0, 1, 2, 3}}, {"quad (1,2,3,4) not 16-byte aligned", {1, 2, 3, 4}},
      {"quad (6,7,8,9) across a 32-byte sector", {6, 7, 8, 9}}, {"quad (14,15,16,17) across the half line", {14, 15, 16, 17}},
      {"quad (30,31,32,33) across two lines", {30, 31, 32, 33}}, {"quad (0,0,0,0)", {0, 0, 0, 0}},
      {"quad (0,0,1,2)", {0, 0, 1, 2}}, {"quad (0,1,1,2)", {0, 1, 1, 2}}, {"quad (0,2,4,6)", {0, 2, 4, 6}},
      {"quad (3,2,1,0)", {3, 2, 1, 0}}, {"quad (0,2,1,3)", {0, 2, 1, 3}}, {"quad (0,1,2,4)", {0, 1, 2, 4}},
      {"quad (0,1,2,8)", {0, 1, 2, 8}}, {"quad (0,1,4,5)", {0, 1, 4, 5}}, {"quad (0,1,8,9)", {0, 1, 8, 9}},
      {"quad (0,4,8,12)", {0, 4, 8, 12}}, {"quad (0,8,16,24)", {0, 8, 16, 24}}, {"quad (0,1,2,7)", {0, 1, 2, 7}},
      {"quad (0,3,5,7) inside one sector", {0, 3, 5, 7}}, {"quad (0,5,10,15) inside the half line", {0, 5, 10, 15}},
      {"quad (0,9,18,27) inside the line", {0, 9, 18, 27}},
    };
    for (const Q& q : qs) {
      const int o0 = q.o[0], o1 = q.o[1], o2 = q.o[2], o3 = q.o[3];
      add(q.name, [=](int k, int l) { const int o[4] = {o0, o1, o2, o3}; return (unsigned)(k * 512 + (l / 4) * 32 + o[l % 4]); }, 64);
    }
    // the same offsets dealt with stride 4 over a 16-lane row: lanes j, j+4, j+8, j+12 of a row read line (row * 4 + j)
    for (const Q& q : qs) {
      const int o0 = q.o[0], o1 = q.o[1], o2 = q.o[2], o3 = q.o[3];
      std::string nm = std::string("stride-4 ") + q.name;
      add(nm.c_str(), [=](int k, int l) { const int o[4] = {o0, o1, o2, o3}; const int r = l / 16, j = l % 4, i = (l % 16) / 4;
                                  return (unsigned)(k * 512 + (r * 4 + j) * 32 + o[i]); }, 64);
    }
    // 16-lane row = one line, arbitrary dwords of it
    add("row of 16 lanes = one line, dword = lane % 16 * 2", [](int k, int l) { return (unsigned)(k * 128 + (l / 16) * 32 + (l % 16) * 2); }, 64);
    add("row of 16 lanes = one line, dword = (lane * 7) % 32", [](int k, int l) { return (unsigned)(k * 128 + (l / 16) * 32 + ((l % 16) * 7) % 32); }, 64);
    add("row of 16 lanes = two lines alternating, consecutive dwords", [](int k, int l) { return (unsigned)(k * 256 + (l / 16) * 64 + (l % 2) * 32 + (l % 16) / 2); }, 64);
    add("row of 16 lanes = 11 + 5 dwords of two lines (window rows)", [](int k, int l) { const int j = l % 16; return (unsigned)(k * 256 + (l / 16) * 64 + (j < 11 ? 3 + j : 32 + 3 + j - 11)); }, 64);
    add("row of 16 lanes = 8 + 3 + 5: tile rows of two tiles and the next row", [](int k, int l) { const int j = l % 16;
         return (unsigned)(k * 512 + (l / 16) * 128 + (j < 5 ? 3 + j : j < 11 ? 32 + j - 5 : 8 + 3 + j - 11)); }, 64);
  }
  // the sweep kernel's pattern: group g = task, lane j = tap j + 16 k of an 11 x 11 window at unit scale on the
  // tiled packed image (8 x 4 entries per line), row-major and column-major dealing
  auto tiled = [](int x, int y) { return (unsigned)(((y >> 2) * 4 + (x >> 3)) * 32 + (y & 3) * 8 + (x & 7)); };  // 4 tiles per row
  add("sweep kernel, row-major taps, 4 tasks at unrelated places", [=](int k, int l) {
    const int g = l / 16, j = l % 16, t = j + 16 * k, tt = t < 121 ? t : 0;
    return tiled(3 + g * 5 % 8 + tt % 11, 1 + g + tt / 11) + (unsigned)g * 1024; }, 32);
  add("sweep kernel, column-major taps, 4 tasks at unrelated places", [=](int k, int l) {
    const int g = l / 16, j = l % 16, t = j + 16 * k, tt = t < 121 ? t : 0;
    return tiled(3 + g * 5 % 8 + tt / 11, 1 + g + tt % 11) + (unsigned)g * 1024; }, 32);
  add("sweep kernel, row-major taps, 4 tasks on adjacent columns", [=](int k, int l) {
    const int g = l / 16, j = l % 16, t = j + 16 * k, tt = t < 121 ? t : 0;
    return tiled(3 + g + tt % 11, 1 + tt / 11); }, 32);

  // 2-byte entries, row-major (pitch 128 halfwords here), tap (tx, ty) of an 11 x 11 window at scale s: entry
  // floor(x0 + tx * s), row floor(y0 + ty * s); offsets in halfwords (meaningful in the halfword-granular section)
  for (float sc : {0.9f, 1.0f, 1.1f, 1.25f}) {
    char nm[128]; snprintf(nm, sizeof nm, "halfword entries row-major, window at scale %.2f, 4 unrelated tasks", sc);
    add(nm, [=](int k, int l) { const int g = l / 16, j = l % 16, t = j + 16 * k, tt = t < 121 ? t : 0;
      const int x = (int)(5.3f + g * 7 + (tt % 11) * sc), y = (int)(1.2f + (tt / 11) * sc);
      return (unsigned)(g * 4096 + y * 128 + x); }, 64);
    snprintf(nm, sizeof nm, "dword entries in 16 x 2 blocks, window at scale %.2f, 4 unrelated tasks", sc);
    add(nm, [=](int k, int l) { const int g = l / 16, j = l % 16, t = j + 16 * k, tt = t < 121 ? t : 0;
      const int x = (int)(5.3f + g * 7 + (tt % 11) * sc), y = (int)(1.2f + (tt / 11) * sc);
      return (unsigned)(g * 2048 + (x >> 4) * 512 + y * 16 + (x & 15)); }, 64);
  }
  unsigned *d_buf, *d_pat, *d_out;
  hipMalloc(&d_buf, region * 4 * 4);
  hipMemset(d_buf, 1, region * 4 * 4);
  hipMalloc(&d_pat, (8 * 64 + 1) * 4);
  hipMalloc(&d_out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int waves_per_cu = 16;
  printf("gfx950 gather rates: %d CUs x %d waves, 8 loads per iteration, %d iterations, clock %.2f GHz\n", cus, waves_per_cu, iters, ghz);
  for (int width : {4, 5}) {
    printf(width == 5 ? "--- 4-byte loads per lane at 2-byte granularity (offsets count halfwords)\n" : "--- %d-byte loads per lane\n", width);
    for (auto& p : pats) {
      hipMemcpy(d_pat, p.off.data(), p.off.size() * 4, hipMemcpyHostToDevice);
      // distinct 128-byte lines / 64-byte halves / 32-byte sectors of load 0 (per instruction and per quad)
      auto count = [&](int gran_dwords, int lanes_per_group) {
        double total = 0;
        for (int g0 = 0; g0 < 64; g0 += lanes_per_group) {
          std::vector<unsigned> s;
          for (int l = g0; l < g0 + lanes_per_group; ++l) {
            const unsigned a = p.off[l] / gran_dwords;
            bool seen = false; for (unsigned q : s) seen |= q == a;
            if (!seen) s.push_back(a);
          }
          total += s.size();
        }
        return total;
      };
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather, dim3(cus * waves_per_cu), dim3(64), 0, 0, d_buf, d_pat, region, iters, d_out, width);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
      }
      const double instr_per_cu = (double)waves_per_cu * iters * 8;
      printf("%-78s %7.1f cycles/instr/CU   lines %2.0f (per quad %4.1f)  halves %2.0f  sectors %2.0f\n", p.name.c_str(),
             best * 1e-3 * ghz * 1e9 / instr_per_cu, count(32, 64), count(32, 4) / 16.0, count(16, 64), count(8, 64));
    }
  }
  return 0;
}
