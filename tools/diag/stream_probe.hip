// stream_probe.hip -- what a CU can stream from HBM in the access shapes of the dense kernels (diagnostic).
// One 512-thread workgroup per CU (LDS-limited, like the dense kernels) reads a contiguous chunk of `rows` x 512 B in
// "images" of IMG_ROWS rows x 256 B (one feature half), DEPTH images in flight, optionally with a barrier per image.
//   pattern 0: lane loads 2 x float4 at (piece * 32 B) and +16 B   (what dense_stage_load does: 8 consecutive floats)
//   pattern 1: lane loads 2 x float4, each instruction lane-contiguous (16 B x 64 lanes = 1 KiB per instruction)
// build: hipcc -O3 --offload-arch=gfx950 tools/diag/stream_probe.hip -o build/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int T = 512;

// STORE: 0 none; 1: one output row of 256 B per 4 lanes x 4 instructions (16 rows x 64 B per instruction: the shape of
// dense_store_acc); 2: lane-contiguous 1 KiB per instruction.  Output volume = 3/4 of the input (dV, dQ, dK vs 4 inputs).
template <int PATTERN, int DEPTH, bool BARRIER, bool FULLROW, int STORE = 0>
__global__ __launch_bounds__(T) void probe(const float *__restrict__ buf, float *__restrict__ out, int rows_per_wg,
                                           int img_rows, float *__restrict__ obuf = nullptr) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const float *base = buf + (size_t)blockIdx.x * rows_per_wg * 128;
  constexpr int PER = 3;           // pieces of 8 floats per thread per image (160 rows x 8 pieces / 512)
  float4 a[DEPTH][PER], b[DEPTH][PER];
  const int nimg = rows_per_wg / img_rows * (FULLROW ? 1 : 2);
  auto fetch = [&](int slot, int q) {
    // image q: rows [r0, r0 + img_rows), feature half hh (or the whole 512-byte row when FULLROW)
    const int r0 = FULLROW ? q * img_rows : (q / 2) * img_rows, hh = FULLROW ? 0 : (q & 1);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int idx = tid + k * T;
      if (PATTERN == 0) {
        const int C8 = FULLROW ? 16 : 8;
        const int row = min(idx / C8, img_rows - 1), c8 = idx % C8;
        const float *p = base + (size_t)(r0 + row) * 128 + hh * 64 + 8 * c8;
        a[slot][k] = *reinterpret_cast<const float4 *>(p);
        b[slot][k] = *reinterpret_cast<const float4 *>(p + 4);
      } else if (PATTERN == 2) {  // pattern 0 with non-temporal loads
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int C8 = FULLROW ? 16 : 8;
        const int row = min(idx / C8, img_rows - 1), c8 = idx % C8;
        const float *p = base + (size_t)(r0 + row) * 128 + hh * 64 + 8 * c8;
        const f4 x = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p)), y = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p + 4));
        a[slot][k] = make_float4(x[0], x[1], x[2], x[3]);
        b[slot][k] = make_float4(y[0], y[1], y[2], y[3]);
      } else {
        // lane-contiguous: instruction k covers float4 chunks idx of the image; the second instruction the other half
        const int C4 = FULLROW ? 32 : 16;  // float4 chunks per row (half)
        const int i0 = min(idx, img_rows * C4 / 2 - 1), i1 = i0 + img_rows * C4 / 2;
        const float *p0 = base + (size_t)(r0 + i0 / C4) * 128 + hh * 64 + 4 * (i0 % C4);
        const float *p1 = base + (size_t)(r0 + i1 / C4) * 128 + hh * 64 + 4 * (i1 % C4);
        a[slot][k] = *reinterpret_cast<const float4 *>(p0);
        b[slot][k] = *reinterpret_cast<const float4 *>(p1);
      }
    }
  };
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) fetch(d, d);
  for (int q0 = 0; q0 < nimg; q0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int q = q0 + d;
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        acc += a[d][k].x + a[d][k].w + b[d][k].y + b[d][k].z;
        lds[(tid + k * T) & 8191] = acc;  // (a use the compiler cannot drop)
      }
      if (q + DEPTH < nimg) fetch(d, q + DEPTH);
      if (STORE != 0 && (q & 3) != 3) {  // 3 of 4 images produce an output half image: 160 rows x 256 B
        float *ob = obuf + (size_t)blockIdx.x * rows_per_wg * 128 + (size_t)((q / 2) * img_rows) * 128 + (q & 1) * 64;
        const int wave = tid >> 6, lane = tid & 63, mi = lane & 15, mq = lane >> 4;
        const float4 v = make_float4(acc, acc, acc, acc);
        for (int strip = wave; strip < img_rows / 16; strip += 8) {
#pragma unroll
          for (int ft = 0; ft < 4; ++ft) {
            if (STORE == 1) {
              *reinterpret_cast<float4 *>(ob + (size_t)(strip * 16 + mi) * 128 + 16 * ft + 4 * mq) = v;
            } else if (STORE == 2) {  // the strip's 16 rows x 256 B as 4 x 1 KiB: instruction ft covers rows 4 ft .. 4 ft + 3 whole
              *reinterpret_cast<float4 *>(ob + (size_t)(strip * 16 + 4 * ft + (lane >> 4)) * 128 + 4 * (lane & 15)) = v;
            } else {  // 3: the same shape, non-temporal stores
              typedef float f4 __attribute__((ext_vector_type(4)));
              const f4 w = {acc, acc, acc, acc};
              __builtin_nontemporal_store(w, reinterpret_cast<f4 *>(ob + (size_t)(strip * 16 + 4 * ft + (lane >> 4)) * 128 + 4 * (lane & 15)));
            }
          }
        }
      }
      if (BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
  if (acc == 123.456f) out[blockIdx.x] = acc + lds[tid];
}

template <int PATTERN, int DEPTH, bool BARRIER, bool FULLROW, int STORE = 0>
static void run(const char *name, const float *buf, float *out, int wgs, int rows_per_wg, int img_rows, float *obuf = nullptr) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<PATTERN, DEPTH, BARRIER, FULLROW, STORE>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it)
    probe<PATTERN, DEPTH, BARRIER, FULLROW, STORE><<<wgs, T, 120 * 1024>>>(buf, out, rows_per_wg, img_rows, obuf);
  hipEventRecord(e0);
  const int reps = 10;
  for (int it = 0; it < reps; ++it)
    probe<PATTERN, DEPTH, BARRIER, FULLROW, STORE><<<wgs, T, 120 * 1024>>>(buf, out, rows_per_wg, img_rows, obuf);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * rows_per_wg * 512.0 * (STORE ? 1.75 : 1.0);
  // PER * T pieces of 32 B are requested per image whatever img_rows is; count the bytes of the rows actually covered
  printf("%-46s %8.1f us  %6.2f TB/s  (%d WGs x %d rows)\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12, wgs,
         rows_per_wg);
}

int main(int argc, char **argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 1024;
  const int rows_per_wg = 1120;  // 7 images of 160 rows (x 2 halves): ~573 KB per workgroup, like a 140-node backward
  const size_t n = (size_t)wgs * rows_per_wg * 128;
  float *buf, *out;
  hipMalloc(&buf, n * 4);
  hipMalloc(&out, wgs * 4);
  hipMemset(buf, 0, n * 4);
  printf("buffer %.1f MB\n", n * 4 / 1e6);
  run<0, 1, true, false>("half rows, 8-float pieces, depth 1, barrier", buf, out, wgs, rows_per_wg, 160);
  run<0, 2, true, false>("half rows, 8-float pieces, depth 2, barrier", buf, out, wgs, rows_per_wg, 160);
  run<0, 3, true, false>("half rows, 8-float pieces, depth 3, barrier", buf, out, wgs, rows_per_wg, 160);
  run<0, 3, false, false>("half rows, 8-float pieces, depth 3, no barrier", buf, out, wgs, rows_per_wg, 160);
  run<1, 1, true, false>("half rows, lane-contiguous, depth 1, barrier", buf, out, wgs, rows_per_wg, 160);
  run<1, 2, true, false>("half rows, lane-contiguous, depth 2, barrier", buf, out, wgs, rows_per_wg, 160);
  run<1, 3, true, false>("half rows, lane-contiguous, depth 3, barrier", buf, out, wgs, rows_per_wg, 160);
  run<1, 3, false, false>("half rows, lane-contiguous, depth 3, no barrier", buf, out, wgs, rows_per_wg, 160);
  float *obuf;
  hipMalloc(&obuf, n * 4);
  run<0, 2, true, false, 1>("half rows, depth 2, + stores 16 rows x 64 B", buf, out, wgs, rows_per_wg, 160, obuf);
  run<0, 2, true, false, 2>("half rows, depth 2, + stores 4 rows x 256 B", buf, out, wgs, rows_per_wg, 160, obuf);
  run<0, 1, true, false, 1>("half rows, depth 1, + stores 16 rows x 64 B", buf, out, wgs, rows_per_wg, 160, obuf);
  run<0, 1, true, false, 2>("half rows, depth 1, + stores 4 rows x 256 B", buf, out, wgs, rows_per_wg, 160, obuf);
  run<0, 1, true, false, 3>("half rows, depth 1, + NT stores 4 rows x 256 B", buf, out, wgs, rows_per_wg, 160, obuf);
  run<2, 1, true, false, 0>("half rows, depth 1, NT loads", buf, out, wgs, rows_per_wg, 160, obuf);
  run<2, 1, true, false, 3>("half rows, depth 1, NT loads + NT stores", buf, out, wgs, rows_per_wg, 160, obuf);
  run<0, 1, true, false, 2>("half rows, depth 1, + stores 4 rows x 256 B (again)", buf, out, wgs, rows_per_wg, 160, obuf);
  run<0, 2, true, true>("full rows (80/img), 8-float pieces, depth 2", buf, out, wgs, rows_per_wg, 80);
  run<1, 2, true, true>("full rows (80/img), lane-contiguous, depth 2", buf, out, wgs, rows_per_wg, 80);
  run<1, 4, false, true>("full rows (80/img), lane-contiguous, depth 4, nb", buf, out, wgs, rows_per_wg, 80);
  return 0;
}
