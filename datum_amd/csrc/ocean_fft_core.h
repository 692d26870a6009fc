// ocean_fft_core.h -- register-resident mixed-radix Stockham inverse FFT of one line.
//
// Replaces the radix-2, log2(N)-barrier LDS ping-pong of data/ocean.fftx.comp:55-99 /
// ocean.ffty.comp:55-99 (one butterfly per thread per stage, two global twiddle loads per
// stage) by at most three passes: every thread keeps E = 4/8/16 points of the line in VGPRs,
// does a radix-E butterfly in registers, and only exchanges through LDS between passes.
// The transform computed is the same one: out[n] = sum_k in[k] exp(+2 pi i k n / N)
// (= N * IFFT, natural order in and out: conj -> forward -> conj of fftx.comp:60-62,97-99).
//
// Line layout while in registers: thread t (0 <= t < T = N/E) holds element t + T*s in slot s,
// both before pass 0 and after the last pass.
//
// The file is host/device neutral so the index arithmetic can be exercised on a CPU by
// emulating the T threads of a line (tests/cpu/fft_core_emul.cpp); kernels live in
// ocean_kernels.hip.

#pragma once

#if defined(__HIPCC__)
#define OC_HD __host__ __device__ __forceinline__
#define OC_UNROLL _Pragma("unroll")
#if defined(__HIP_DEVICE_COMPILE__)
#define OC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define OC_SCHED_FENCE() do { } while(0)
#endif
#else
#define OC_SCHED_FENCE() do { } while(0)
#define OC_HD inline
#define OC_UNROLL
#endif

namespace ocean
{
  struct alignas(8) cf
  {
    float x, y;
  };

  OC_HD cf operator+(cf a, cf b) { return { a.x + b.x, a.y + b.y }; }
  OC_HD cf operator-(cf a, cf b) { return { a.x - b.x, a.y - b.y }; }

  OC_HD float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

#if defined(__HIP_DEVICE_COMPILE__)
  // Packed-fp32 complex arithmetic with the half-select / negate operand modifiers of VOP3P written out.
  // hipcc materialises the swapped or negated operand with extra v_mov / v_xor instead (4 instructions per
  // complex multiply, 3-4 per "+- i*d"); these are 2 and 1.  Same roundings as the portable forms below.
  typedef float v2f_ __attribute__((ext_vector_type(2)));

  // a * b = (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
  __device__ __forceinline__ cf cmul(cf a, cf b)
  {
    v2f_ va = __builtin_bit_cast(v2f_, a), vb = __builtin_bit_cast(v2f_, b), t, r;

    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(va), "v"(vb));                                  // (a.y b.y, a.y b.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(va), "v"(vb), "v"(t));                   // (a.x b.x - t.x, a.x b.y + t.y)

    return __builtin_bit_cast(cf, r);
  }

  // a + i*d = (a.x - d.y, a.y + d.x)   and   a - i*d = (a.x + d.y, a.y - d.x)
  __device__ __forceinline__ cf add_muli(cf a, cf d)
  {
    v2f_ va = __builtin_bit_cast(v2f_, a), vd = __builtin_bit_cast(v2f_, d), r;

    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(va), "v"(vd));

    return __builtin_bit_cast(cf, r);
  }

  __device__ __forceinline__ cf sub_muli(cf a, cf d)
  {
    v2f_ va = __builtin_bit_cast(v2f_, a), vd = __builtin_bit_cast(v2f_, d), r;

    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(va), "v"(vd));

    return __builtin_bit_cast(cf, r);
  }

  // a + h*b and a - h*b for a real constant h (one packed FMA each)
  __device__ __forceinline__ cf fma_real(cf a, cf b, float h)
  {
    v2f_ va = __builtin_bit_cast(v2f_, a), vb = __builtin_bit_cast(v2f_, b), vh = { h, h }, r;

    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(vb), "v"(vh), "v"(va));

    return __builtin_bit_cast(cf, r);
  }

  __device__ __forceinline__ cf fms_real(cf a, cf b, float h)
  {
    v2f_ va = __builtin_bit_cast(v2f_, a), vb = __builtin_bit_cast(v2f_, b), vh = { h, h }, r;

    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(vb), "v"(vh), "v"(va));

    return __builtin_bit_cast(cf, r);
  }

  // a + conj(b) = (a.x + b.x, a.y - b.y)
  __device__ __forceinline__ cf add_conj(cf a, cf b)
  {
    v2f_ va = __builtin_bit_cast(v2f_, a), vb = __builtin_bit_cast(v2f_, b), r;

    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(va), "v"(vb));

    return __builtin_bit_cast(cf, r);
  }

  // a + c * conj(b) for a real c = (a.x + c b.x, a.y - c b.y)
  __device__ __forceinline__ cf fma_conj(cf a, cf b, float c)
  {
    v2f_ va = __builtin_bit_cast(v2f_, a), vb = __builtin_bit_cast(v2f_, b), vc = { c, c }, r;

    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_hi:[1,0,0]" : "=v"(r) : "v"(vb), "v"(vc), "v"(va));

    return __builtin_bit_cast(cf, r);
  }

  // a + c * (-i b) for a real c = (a.x + c b.y, a.y - c b.x)
  __device__ __forceinline__ cf fma_negi(cf a, cf b, float c)
  {
    v2f_ va = __builtin_bit_cast(v2f_, a), vb = __builtin_bit_cast(v2f_, b), vc = { c, c }, r;

    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(vb), "v"(vc), "v"(va));

    return __builtin_bit_cast(cf, r);
  }

  // c * a for a real c
  __device__ __forceinline__ cf scale_real(cf a, float c)
  {
    v2f_ va = __builtin_bit_cast(v2f_, a), vc = { c, c }, r;

    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(va), "v"(vc));

    return __builtin_bit_cast(cf, r);
  }
#else
  // a * b
  OC_HD cf cmul(cf a, cf b) { return { fmaf_(a.x, b.x, -(a.y * b.y)), fmaf_(a.x, b.y, a.y * b.x) }; }

  OC_HD cf add_conj(cf a, cf b) { return { a.x + b.x, a.y - b.y }; }
  OC_HD cf fma_conj(cf a, cf b, float c) { return { fmaf_(c, b.x, a.x), fmaf_(-c, b.y, a.y) }; }
  OC_HD cf fma_negi(cf a, cf b, float c) { return { fmaf_(c, b.y, a.x), fmaf_(-c, b.x, a.y) }; }
  OC_HD cf scale_real(cf a, float c) { return { c * a.x, c * a.y }; }

  OC_HD cf fma_real(cf a, cf b, float h) { return { fmaf_(h, b.x, a.x), fmaf_(h, b.y, a.y) }; }
  OC_HD cf fms_real(cf a, cf b, float h) { return { fmaf_(-h, b.x, a.x), fmaf_(-h, b.y, a.y) }; }

  OC_HD cf add_muli(cf a, cf d) { return { a.x - d.y, a.y + d.x }; }
  OC_HD cf sub_muli(cf a, cf d) { return { a.x + d.y, a.y - d.x }; }
#endif

  // i * a
  OC_HD cf muli(cf a) { return { -a.y, a.x }; }

  //|---------------------- plan ----------------------------------------------

  // (a loop, not a recursion: where the exponent is an unrolled loop counter rather than a constant expression the call
  // must inline and fold; hipcc emitted a real device-function call plus runtime divisions for the recursive form in
  // one translation unit -- 155 instead of 100 VGPRs in the row pass -- and folded it in another)
  OC_HD constexpr int ipow(int b, int e)
  {
    int r = 1;

    for(int i = 0; i < e; ++i)
      r *= b;

    return r;
  }

  // N = E^(NP-1) * RL: NP-1 passes of radix E, one last pass of radix RL <= E done as M = E/RL tasks.
  // E = 8 keeps a line transform near 50 VGPRs (E = 16: ~90) so that several lines per thread fit in registers.
  constexpr int plan_passes(int n, int e) { return n <= e ? 1 : 1 + plan_passes(n / e, e); }

  // points per thread of a line transform unless the caller chooses (the kernels do, per pass: RowCfg / ColCfg)
  constexpr int default_radix(int n) { return n == 64 ? 4 : 8; }

  template<int N, int E_ = default_radix(N)>
  struct Plan
  {
    static_assert(N == 64 || N == 128 || N == 256 || N == 512 || N == 1024 || N == 2048 || N == 4096, "unsupported resolution");

    static constexpr int E = E_;
    static constexpr int T = N / E;
    static constexpr int NP = plan_passes(N, E);
    static constexpr int RL = N / ipow(E, NP - 1);
    static constexpr int M = E / RL;
    static constexpr int NS_LAST = N / RL;      // product of the radices before the last pass
    static_assert(NP >= 2 && NP <= 6, "bad plan");
    static_assert(RL >= 2 && RL <= E && E % RL == 0, "bad plan");
    static_assert(ipow(E, NP - 1) * RL == N, "bad plan");
  };

  //|---------------------- LDS layout of the exchanges ----------------------
  // Every exchange between two passes has its own layout, chosen so that BOTH its sides are free of LDS bank conflicts
  // on gfx950 and cost no address arithmetic (a per-thread base plus compile-time offsets):
  //   ds_read_b64 is serviced in two groups of 32 lanes over 64 banks of 4 bytes, ds_write_b64 in four groups of 16 lanes
  //   over 32 banks (MI355X_MICROARCH.md, LDS) -- 32 lanes must read 32 elements with 32 different positions mod 32,
  //   16 lanes must write 16 elements with 16 different positions mod 16.
  // A pass's loads are always lane-contiguous in the element index (t + T r: the autosort property), its stores are not.
  // W = lines interleaved element by element (the column pass's W columns: threads are column-fastest, address =
  // position * W + column, so that consecutive LANES touch consecutive addresses); W = 1 in the row pass.
  //   behind pass 0 (stores E t + q): TRANSPOSED, position = q (T + PAD0) + t.  Stores: consecutive lanes, consecutive
  //       addresses.  Loads of element t' + T r = E a + b: position b (T + PAD0) + a with W (T + PAD0) = 4 mod 8 (E = 8) or
  //       2 mod 4 (E = 16), which spreads the lanes' b over the banks that the consecutive a leave free.
  //   behind middle pass P (Ns = E^P, stores (t / Ns) Ns E + t % Ns + q Ns): the element index itself, plus Ns elements
  //       of padding per Ns E where 16 lanes span more than one run of Ns (Ns W < 16: only the row pass's radix-8 plan)
  // Round 4's i + (i >> 4) / i + (i >> 3) paddings put a gap inside every run of 32 consecutive elements: every
  // ds_read_b64 of the transforms took two LDS cycles per lane group instead of one (tools/lds/bank_model.py reproduces the
  // measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of 21-48 % per kernel and gives 0 for this layout from 256^2 up).
  // The kernels are compiled with hipcc's pairing of LDS accesses switched off (OCEAN_LDS_UNPAIRED on the kernel: the
  // "load-store-opt" target feature): two ds_read_b64 whose addresses differ by a constant otherwise become one ds_read2_b64 /
  // ds_read2st64_b64, which the LDS serves in 16-lane groups over 32 banks at 128 bytes per clock -- half the rate of the two
  // ds_read_b64 it replaces (measured: SQ_LDS_IDX_ACTIVE 17.1 M against 11.9 M per launch of the 4096^2 column pass, and 12 % of
  // conflict cycles back: profiles/r05_lds_conflicts.txt).
  constexpr int lds_pad0(int e, int w) { return ((32 / e) / w) > 1 ? (32 / e) / w : 1; }

  //|---------------------- radix butterflies ---------------------------------
  // idft<R>: v[q] <- sum_r v[r] exp(+2 pi i q r / R), natural order

  OC_HD void idft2(cf &a, cf &b)
  {
    cf s = a + b, d = a - b;
    a = s;
    b = d;
  }

  OC_HD void idft4(cf &a0, cf &a1, cf &a2, cf &a3)
  {
    cf s02 = a0 + a2, d02 = a0 - a2;
    cf s13 = a1 + a3, d13 = a1 - a3;

    a0 = s02 + s13;
    a1 = add_muli(d02, d13);
    a2 = s02 - s13;
    a3 = sub_muli(d02, d13);
  }

  template<int R> struct Radix;

  template<> struct Radix<2>
  {
    static OC_HD void run(cf (&v)[2]) { idft2(v[0], v[1]); }
  };

  template<> struct Radix<4>
  {
    static OC_HD void run(cf (&v)[4]) { idft4(v[0], v[1], v[2], v[3]); }
  };

  template<> struct Radix<8>
  {
    static OC_HD void run(cf (&v)[8])
    {
      const float h = 0.70710678118654752440f;

      // index = a + 2b: 4-point transforms over b for a = 0, 1
      idft4(v[0], v[2], v[4], v[6]);
      idft4(v[1], v[3], v[5], v[7]);

      // t[1][c] *= exp(2 pi i c / 8):  t1 = h (1 + i) v3,  t3 = -h (1 - i) v7, the factor h folded into the sums
      cf s1 = add_muli(v[3], v[3]);
      cf s3 = sub_muli(v[7], v[7]);

      cf o[8];
      o[0] = v[0] + v[1];  o[4] = v[0] - v[1];
      o[1] = fma_real(v[2], s1, h);   o[5] = fms_real(v[2], s1, h);
      o[2] = add_muli(v[4], v[5]);    o[6] = sub_muli(v[4], v[5]);
      o[3] = fms_real(v[6], s3, h);   o[7] = fma_real(v[6], s3, h);

      OC_UNROLL
      for(int i = 0; i < 8; ++i)
        v[i] = o[i];
    }
  };

  template<> struct Radix<16>
  {
    static OC_HD void run(cf (&v)[16])
    {
      const float h = 0.70710678118654752440f;
      const float c1 = 0.92387953251128675613f;   // cos(pi/8)
      const float s1 = 0.38268343236508977173f;   // sin(pi/8)

      // index = a + 4b: 4-point transforms over b for a = 0..3; slot a + 4c then holds t[a][c]
      idft4(v[0], v[4], v[8], v[12]);
      idft4(v[1], v[5], v[9], v[13]);
      idft4(v[2], v[6], v[10], v[14]);
      idft4(v[3], v[7], v[11], v[15]);

      // t[a][c] *= exp(2 pi i a c / 16)
      v[5] = cmul(v[5], cf{ c1, s1 });                                          // a=1 c=1 : w^1
      v[9] = cf{ (v[9].x - v[9].y) * h, (v[9].x + v[9].y) * h };                // a=1 c=2 : w^2
      v[13] = cmul(v[13], cf{ s1, c1 });                                        // a=1 c=3 : w^3
      v[6] = cf{ (v[6].x - v[6].y) * h, (v[6].x + v[6].y) * h };                // a=2 c=1 : w^2
      v[10] = muli(v[10]);                                                      // a=2 c=2 : w^4
      v[14] = cf{ (-v[14].x - v[14].y) * h, (v[14].x - v[14].y) * h };          // a=2 c=3 : w^6
      v[7] = cmul(v[7], cf{ s1, c1 });                                          // a=3 c=1 : w^3
      v[11] = cf{ (-v[11].x - v[11].y) * h, (v[11].x - v[11].y) * h };          // a=3 c=2 : w^6
      v[15] = cmul(v[15], cf{ -c1, -s1 });                                      // a=3 c=3 : w^9

      // 4-point transforms over a for each c; output q = c + 4d
      cf o[16];
      OC_UNROLL
      for(int c = 0; c < 4; ++c)
      {
        cf x0 = v[4*c+0], x1 = v[4*c+1], x2 = v[4*c+2], x3 = v[4*c+3];
        idft4(x0, x1, x2, x3);
        o[c] = x0;
        o[c+4] = x1;
        o[c+8] = x2;
        o[c+12] = x3;
      }

      OC_UNROLL
      for(int i = 0; i < 16; ++i)
        v[i] = o[i];
    }
  };

  // 8-point transform of v[r] w^r with the twiddle w^r = w^a (w^2)^b of index r = a + 2b applied in two steps (see idft16_twiddled):
  // four twiddle values live (w, w^2, w^4, w^6) instead of seven powers, the same 13 complex products (3 + 10 against 6 + 7)
  OC_HD void idft8_twiddled(cf (&v)[8], cf w1)
  {
    const float h = 0.70710678118654752440f;

    {
      cf const w2 = cmul(w1, w1);
      cf const w4 = cmul(w2, w2);
      cf const w6 = cmul(w4, w2);

      v[2] = cmul(v[2], w2);  v[3] = cmul(v[3], w2);
      v[4] = cmul(v[4], w4);  v[5] = cmul(v[5], w4);
      v[6] = cmul(v[6], w6);  v[7] = cmul(v[7], w6);
    }

    // index = a + 2b: 4-point transforms over b for a = 0, 1
    idft4(v[0], v[2], v[4], v[6]);
    idft4(v[1], v[3], v[5], v[7]);

    // ... and by w^a
    v[1] = cmul(v[1], w1);
    v[3] = cmul(v[3], w1);
    v[5] = cmul(v[5], w1);
    v[7] = cmul(v[7], w1);

    // t[1][c] *= exp(2 pi i c / 8):  t1 = h (1 + i) v3,  t3 = -h (1 - i) v7, the factor h folded into the sums
    cf s1 = add_muli(v[3], v[3]);
    cf s3 = sub_muli(v[7], v[7]);

    cf o[8];
    o[0] = v[0] + v[1];  o[4] = v[0] - v[1];
    o[1] = fma_real(v[2], s1, h);   o[5] = fms_real(v[2], s1, h);
    o[2] = add_muli(v[4], v[5]);    o[6] = sub_muli(v[4], v[5]);
    o[3] = fms_real(v[6], s3, h);   o[7] = fma_real(v[6], s3, h);

    OC_UNROLL
    for(int i = 0; i < 8; ++i)
      v[i] = o[i];
  }

  // 16-point transform of v[r] w^r (the last pass of 4096 = 16^3): the twiddle w^r = w^a (w^4)^b of index r = a + 4b is applied in
  // two steps, (w^4)^b in front of the 4-point transforms over b and w^a behind them -- six twiddle values live (w, w^2, w^3 and
  // w^4, w^8, w^12) instead of the fifteen powers of a product tree (thirty registers beside the thirty-two of the values: the
  // sequential row pass of 4096^2 spilled them), for the same 29 complex products (5 to form them + 24, against 14 + 15).
  OC_HD void idft16_twiddled(cf (&v)[16], cf w1)
  {
    const float h = 0.70710678118654752440f;
    const float c1 = 0.92387953251128675613f;   // cos(pi/8)
    const float s1 = 0.38268343236508977173f;   // sin(pi/8)

    cf const w2 = cmul(w1, w1);
    cf const w4 = cmul(w2, w2);

    {
      cf const w8 = cmul(w4, w4);
      cf const w12 = cmul(w8, w4);

      OC_UNROLL
      for(int a = 0; a < 4; ++a)
      {
        v[a + 4] = cmul(v[a + 4], w4);
        v[a + 8] = cmul(v[a + 8], w8);
        v[a + 12] = cmul(v[a + 12], w12);
      }
    }

    // index = a + 4b: 4-point transforms over b for a = 0..3; slot a + 4c then holds t[a][c]
    idft4(v[0], v[4], v[8], v[12]);
    idft4(v[1], v[5], v[9], v[13]);
    idft4(v[2], v[6], v[10], v[14]);
    idft4(v[3], v[7], v[11], v[15]);

    // t[a][c] *= exp(2 pi i a c / 16)
    v[5] = cmul(v[5], cf{ c1, s1 });                                          // a=1 c=1 : w^1
    v[9] = cf{ (v[9].x - v[9].y) * h, (v[9].x + v[9].y) * h };                // a=1 c=2 : w^2
    v[13] = cmul(v[13], cf{ s1, c1 });                                        // a=1 c=3 : w^3
    v[6] = cf{ (v[6].x - v[6].y) * h, (v[6].x + v[6].y) * h };                // a=2 c=1 : w^2
    v[10] = muli(v[10]);                                                      // a=2 c=2 : w^4
    v[14] = cf{ (-v[14].x - v[14].y) * h, (v[14].x - v[14].y) * h };          // a=2 c=3 : w^6
    v[7] = cmul(v[7], cf{ s1, c1 });                                          // a=3 c=1 : w^3
    v[11] = cf{ (-v[11].x - v[11].y) * h, (v[11].x - v[11].y) * h };          // a=3 c=2 : w^6
    v[15] = cmul(v[15], cf{ -c1, -s1 });                                      // a=3 c=3 : w^9

    // ... and by w^a
    {
      cf const w3 = cmul(w2, w1);

      OC_UNROLL
      for(int c = 0; c < 4; ++c)
      {
        v[1 + 4 * c] = cmul(v[1 + 4 * c], w1);
        v[2 + 4 * c] = cmul(v[2 + 4 * c], w2);
        v[3 + 4 * c] = cmul(v[3 + 4 * c], w3);
      }
    }

    // 4-point transforms over a for each c; output q = c + 4d
    cf o[16];
    OC_UNROLL
    for(int c = 0; c < 4; ++c)
    {
      cf x0 = v[4*c+0], x1 = v[4*c+1], x2 = v[4*c+2], x3 = v[4*c+3];
      idft4(x0, x1, x2, x3);
      o[c] = x0;
      o[c+4] = x1;
      o[c+8] = x2;
      o[c+12] = x3;
    }

    OC_UNROLL
    for(int i = 0; i < 16; ++i)
      v[i] = o[i];
  }

  // powers w^1 .. w^(R-1) of a unit twiddle by a product tree of depth <= 4
  template<int R>
  OC_HD void twiddle_powers(cf w1, cf (&w)[R])
  {
    w[0] = cf{ 1.0f, 0.0f };
    w[1] = w1;
    if (R > 2)
    {
      w[2] = cmul(w1, w1);
      w[3] = cmul(w[2], w1);
    }
    if (R > 4)
    {
      w[4] = cmul(w[2], w[2]);
      w[5] = cmul(w[4], w1);
      w[6] = cmul(w[4], w[2]);
      w[7] = cmul(w[4], w[3]);
    }
    if (R > 8)
    {
      w[8] = cmul(w[4], w[4]);
      OC_UNROLL
      for(int i = 1; i < 8; ++i)
        w[8+i] = cmul(w[8], w[i]);
    }
  }

  // u[q] <- sum_r u[r] w^r exp(2 pi i q r / R)
  template<int R> struct TwiddledDft
  {
    static OC_HD void run(cf (&u)[R], cf w1)
    {
      cf p[R];
      twiddle_powers<R>(w1, p);

      OC_UNROLL
      for(int r = 1; r < R; ++r)
        u[r] = cmul(u[r], p[r]);

      Radix<R>::run(u);
    }
  };

  template<> struct TwiddledDft<8>
  {
    static OC_HD void run(cf (&u)[8], cf w1) { idft8_twiddled(u, w1); }
  };

  template<> struct TwiddledDft<16>
  {
    static OC_HD void run(cf (&u)[16], cf w1) { idft16_twiddled(u, w1); }
  };

  template<int R> OC_HD void twiddled_dft(cf (&u)[R], cf w1) { TwiddledDft<R>::run(u, w1); }

  //|---------------------- one line ------------------------------------------
  // tw[k] = exp(+2 pi i k / N), k < N   (built in double on the host: ocean_capi)
  // `line` points at this line's padded LDS region (Plan::LINE elements)

  // per-thread twiddles kept in registers, each the first power of its pass; higher powers come from
  // twiddle_powers.  last[m] = exp(2 pi i j / N) for the last pass's tasks j = t + T m (Ns = N / RL);
  // mid[k] = exp(2 pi i (t % Ns) / (Ns E)) for middle pass k + 2 (Ns = E^(k+2)).
  template<int N, int E_ = default_radix(N)>
  struct LineTwiddles
  {
    static constexpr int NMIDREG = (Plan<N, E_>::NP > 3) ? Plan<N, E_>::NP - 3 : 1;

    cf mid[NMIDREG];
    cf last[Plan<N, E_>::M];
  };

  template<int N, int W = 1, int E_ = default_radix(N)>
  struct LineFFT
  {
    typedef Plan<N, E_> P;

    static constexpr int E = P::E;
    static constexpr int T = P::T;
    static constexpr int RL = P::RL;
    static constexpr int M = P::M;

#ifdef OCEAN_EXP_NO_FENCE
    static constexpr bool FENCE_LAST_TASKS = false;
#else
    static constexpr bool FENCE_LAST_TASKS = (N == 2048 && W == 4 && E_ == 16);
#endif

    // exchange 0: transposed, TP elements per q
    static constexpr int TP = T + lds_pad0(E, W);

    static OC_HD constexpr int x0(int i) { return (i % E) * TP + i / E; }

    // exchange behind middle pass PASS: Ns elements of padding per Ns E where 16 lanes span more than one run of Ns
    template<int PASS> static constexpr bool padded() { return ipow(E, PASS) * W < 16; }

    template<int PASS> static OC_HD constexpr int xm(int i)
    {
      constexpr int Ns = ipow(E, PASS);

      return padded<PASS>() ? i + Ns * (i / (Ns * E)) : i;
    }

    // the layout a pass reads: exchange PASS - 1
    template<int PASS> static OC_HD constexpr int xin(int i) { return PASS == 1 ? x0(i) : xm<(PASS > 1 ? PASS - 1 : 1)>(i); }

    static constexpr bool ANYPAD = (P::NP >= 3) && padded<1>();

    // elements per line (of one column); the Hermitian swap of the row pass also fits: N + 1 <= LINE
    static constexpr int LINE = (E * TP > (ANYPAD ? N + N / E : N + 2)) ? E * TP : (ANYPAD ? N + N / E : N + 2);

    // per-thread twiddles kept in registers, each the first power of its pass; higher powers come from
    // twiddle_powers.  last[m] = exp(2 pi i j / N) for the last pass's tasks j = t + T m (Ns = N / RL);
    // mid[k] = exp(2 pi i (t % Ns) / (Ns E)) for middle pass k + 2 (Ns = E^(k+2)).
    static constexpr int NMIDREG = LineTwiddles<N, E_>::NMIDREG;

    typedef LineTwiddles<N, E_> Twiddles;

    static OC_HD void load_twiddles(cf const *tw, int t, Twiddles &w)
    {
      OC_UNROLL
      for(int k = 0; k < NMIDREG; ++k)
      {
        int ns = ipow(E, k + 2);
        w.mid[k] = (P::NP > 3) ? tw[((t % ns) * (N / (ns * E))) % N] : cf{ 1.0f, 0.0f };
      }

      OC_UNROLL
      for(int m = 0; m < M; ++m)
        w.last[m] = tw[((t + T * m) % P::NS_LAST) * (N / (P::NS_LAST * RL))];
    }

    // first middle pass (Ns = E) twiddles, shared by every line: midtab[r E + a] = exp(2 pi i a r / E^2),
    // a = t % E fastest so that a wave reads consecutive entries per r.  E*E entries (in LDS).
    static constexpr int MIDTAB = (P::NP >= 3) ? E * E : 0;

    static OC_HD cf midtab_entry(cf const *tw, int i)
    {
      int a = i % E, r = i / E;

      return tw[(a * r * (N / (E * E))) % N];
    }

    // `line` points at element 0 of this line (column); element positions are W apart

    // pass 0: Ns = 1, radix E, task j = t.  v[s] = x[t + T s] -> element t E + q
    static OC_HD void pass0(cf (&v)[E], int t, cf *line)
    {
      Radix<E>::run(v);

      OC_UNROLL
      for(int q = 0; q < E; ++q)
        line[(q * TP + t) * W] = v[q];
    }

    // element t + T r of the layout pass PASS reads, r a compile-time constant: a per-thread base plus a constant
    template<int PASS> static OC_HD int load_pos(int t, int r)
    {
      if (PASS == 1)
      {
        if (T % E == 0)
          return (t % E) * TP + t / E + (T / E) * r;

        return x0(t + T * r);
      }

      constexpr int Ns = ipow(E, PASS > 1 ? PASS - 1 : 1);

      if (padded<(PASS > 1 ? PASS - 1 : 1)>() && T % (Ns * E) == 0)
        return t + Ns * (t / (Ns * E)) + (T + T / E) * r;

      return xin<PASS>(t + T * r);
    }

    // middle pass PASS (1 <= PASS <= NP-2): Ns = E^PASS, radix E, task j = t.  load + twiddle + butterfly
    template<int PASS>
    static OC_HD void mid_load(cf (&v)[E], int t, cf const *line, cf const *midtab, Twiddles const &w)
    {
      OC_UNROLL
      for(int r = 0; r < E; ++r)
        v[r] = line[load_pos<PASS>(t, r) * W];

      if (PASS == 1)
      {
        OC_UNROLL
        for(int r = 1; r < E; ++r)
          v[r] = cmul(v[r], midtab[r * E + (t % E)]);
      }
      else
      {
        constexpr int MI = (PASS >= 2 && PASS - 2 < NMIDREG) ? PASS - 2 : 0;

        twiddled_dft<E>(v, w.mid[MI]);

        return;
      }

      Radix<E>::run(v);
    }

    template<int PASS>
    static OC_HD void mid_store(cf const (&v)[E], int t, cf *line)
    {
      constexpr int Ns = ipow(E, PASS);

      // element (t / Ns) Ns E + t % Ns + q Ns; its run of Ns E elements starts (t / Ns) Ns (E + 1) on where padded
      int base = (t / Ns) * Ns * (padded<PASS>() ? E + 1 : E) + (t % Ns);

      OC_UNROLL
      for(int q = 0; q < E; ++q)
        line[(base + q * Ns) * W] = v[q];
    }

    // last pass: Ns = N/RL, radix RL, tasks j = t + T m.  result slot m + q M holds X[t + T (m + q M)]
    static OC_HD void last(cf (&v)[E], int t, cf const *line, Twiddles const &w)
    {
      constexpr int LP = P::NP - 1;       // reads what the pass before it stored

      OC_UNROLL
      for(int m = 0; m < M; ++m)
      {
        cf u[RL];
        OC_UNROLL
        for(int r = 0; r < RL; ++r)
          u[r] = line[load_pos<LP>(t, m + r * (N / RL / T)) * W];     // element t + T m + r N / RL

        twiddled_dft<RL>(u, w.last[m]);

        OC_UNROLL
        for(int q = 0; q < RL; ++q)
          v[m + q * M] = u[q];

        // the tasks of the last pass one after the other: hipcc otherwise starts task m + 1's LDS reads and twiddle powers under the tail of
        // task m, and in the one kernel that sits at its register limit -- the 2048^2 column pass: 16 points per thread twice, two 512-thread
        // tiles per CU = 128 registers -- two of task m's results went to scratch for it (20 bytes per lane through round 5)
        if (FENCE_LAST_TASKS && m + 1 < M)
          OC_SCHED_FENCE();
      }
    }
  };

  // element a thread holds in slot s before / after a line transform
  template<int N, int E_ = default_radix(N)> OC_HD constexpr int elem_in(int t, int s) { return t + Plan<N, E_>::T * s; }
  template<int N, int E_ = default_radix(N)> OC_HD constexpr int elem_out(int t, int s) { return t + Plan<N, E_>::T * s; }
}
