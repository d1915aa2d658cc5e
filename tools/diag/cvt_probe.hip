// cvt_probe.hip -- cycles per 8-element fp32 -> fp16 hi/lo split, by code shape (diagnostic; 2 waves per SIMD like the
// dense kernels).  build: hipcc -O3 --offload-arch=gfx950 tools/diag/cvt_probe.hip -o build/cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h16;
typedef __attribute__((ext_vector_type(8))) _Float16 hx8;
typedef __attribute__((ext_vector_type(2))) _Float16 hx2;
typedef __attribute__((ext_vector_type(2))) float f2;

// 0: plain C (the compiler picks v_fma_mix*)
__device__ __forceinline__ void split0(const float (&x)[8], float s, hx8 &hi, hx8 &lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const h16 h = (h16)(x[j] * s);
    hi[j] = h;
    lo[j] = (h16)fmaf(x[j], s, -(float)h);
  }
}
// 1: packed converts, the subtraction in fp32 (fusion into v_fma_mix blocked)
__device__ __forceinline__ void split1(const float (&x)[8], float s, hx8 &hi, hx8 &lo) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    float y0 = x[j] * s, y1 = x[j + 1] * s;
    hx2 h = __builtin_convertvector(f2{y0, y1}, hx2);
    asm volatile("" : "+v"(h));
    f2 hf = __builtin_convertvector(h, f2);
    asm volatile("" : "+v"(hf));
    hx2 l = __builtin_convertvector(f2{y0 - hf.x, y1 - hf.y}, hx2);
    hi[j] = h.x; hi[j + 1] = h.y;
    lo[j] = l.x; lo[j + 1] = l.y;
  }
}
// 2: hi by integer rounding of the fp32 bits to 11 significant bits (exactly representable in fp16), lo = y - hi
__device__ __forceinline__ void split2(const float (&x)[8], float s, hx8 &hi, hx8 &lo) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    float y0 = x[j] * s, y1 = x[j + 1] * s;
    const unsigned u0 = __float_as_uint(y0), u1 = __float_as_uint(y1);
    const float h0 = __uint_as_float((u0 + 0x1000u) & 0xFFFFE000u), h1 = __uint_as_float((u1 + 0x1000u) & 0xFFFFE000u);
    hx2 h = __builtin_convertvector(f2{h0, h1}, hx2);
    hx2 l = __builtin_convertvector(f2{y0 - h0, y1 - h1}, hx2);
    hi[j] = h.x; hi[j + 1] = h.y;
    lo[j] = l.x; lo[j + 1] = l.y;
  }
}
// 3: bf16 split as before (reference point)
typedef __attribute__((ext_vector_type(8))) __bf16 bx8;
__device__ __forceinline__ void split3(const float (&x)[8], float s, bx8 &hi, bx8 &lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)x[j];
    hi[j] = h;
    lo[j] = (__bf16)(x[j] - (float)h);
  }
}

template <int V>
__global__ __launch_bounds__(512) void k(const float *in, float *out, long long *cyc, float s, int iters) {
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = in[threadIdx.x * 8 + j];
  float acc = 0.f;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (V == 3) {
      bx8 h, l;
      split3(x, s, h, l);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += (float)h[j] + (float)l[j];
    } else {
      hx8 h, l;
      if constexpr (V == 0) split0(x, s, h, l);
      if constexpr (V == 1) split1(x, s, h, l);
      if constexpr (V == 2) split2(x, s, h, l);
      // consume without converting back: reinterpret the packed halves as floats and xor them in
      typedef __attribute__((ext_vector_type(4))) unsigned u4;
      const u4 a = __builtin_bit_cast(u4, h), b = __builtin_bit_cast(u4, l);
      const unsigned m = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
      acc = __uint_as_float(__float_as_uint(acc) ^ m);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = __uint_as_float(__float_as_uint(x[j]) + 1u);  // new operands every round
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float *in, *out;
  long long *cyc;
  hipMalloc(&in, 512 * 8 * 4);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 8);
  hipMemset(in, 0x3c, 512 * 8 * 4);
  const int iters = 2000;
  auto run = [&](auto kern, const char *name) {
    long long h[256];
    for (int r = 0; r < 2; ++r) {
      kern<<<256, 512>>>(in, out, cyc, 16384.f, iters);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %.1f cycles per 8-element split per wave (2 waves per SIMD; incl. ~9 ops of loop overhead)\n", name,
           (double)h[0] / iters);
  };
  run(k<0>, "0 plain C (v_fma_mix*)");
  run(k<1>, "1 cvt_pk_f16 + cvt_f32_f16 + sub + cvt_pk");
  run(k<2>, "2 integer-rounded hi + sub + 2 cvt_pk");
  run(k<3>, "3 bf16 hi/lo (round 1)");
  return 0;
}
