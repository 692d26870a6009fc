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
#else
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

#ifndef OCEAN_ASM_COMPLEX
#define OCEAN_ASM_COMPLEX 1
#endif

#if defined(__HIP_DEVICE_COMPILE__) && OCEAN_ASM_COMPLEX
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
#else
  // a * b
  OC_HD cf cmul(cf a, cf b) { return { fmaf_(a.x, b.x, -(a.y * b.y)), fmaf_(a.x, b.y, a.y * b.x) }; }

  OC_HD cf add_muli(cf a, cf d) { return { a.x - d.y, a.y + d.x }; }
  OC_HD cf sub_muli(cf a, cf d) { return { a.x + d.y, a.y - d.x }; }
#endif

  // i * a
  OC_HD cf muli(cf a) { return { -a.y, a.x }; }

  //|---------------------- plan ----------------------------------------------

  constexpr int ipow(int b, int e) { return e == 0 ? 1 : b * ipow(b, e - 1); }

  // N = E^(NP-1) * RL: NP-1 passes of radix E, one last pass of radix RL <= E done as M = E/RL tasks.
  // E = 8 keeps a line transform near 50 VGPRs (E = 16: ~90) so that several lines per thread fit in registers.
  constexpr int plan_passes(int n, int e) { return n <= e ? 1 : 1 + plan_passes(n / e, e); }

#ifndef OCEAN_FFT_E
#define OCEAN_FFT_E 8
#endif
#ifndef OCEAN_QUAD_FFT
#define OCEAN_QUAD_FFT 0      // measured on MI355X: correct, but 30 % more VALU instructions and 50 % slower (profiles/README.md)
#endif

  // N = 1024 runs as 32 x 32 with quad (DPP) butterflies and a single LDS exchange: see the end of this file
  template<int N> struct QuadFFT { static constexpr bool ENABLED = (N == 1024) && (OCEAN_FFT_E == 8) && (OCEAN_QUAD_FFT != 0); };

  template<int N>
  struct Plan
  {
    static_assert(N == 64 || N == 128 || N == 256 || N == 512 || N == 1024 || N == 2048 || N == 4096, "unsupported resolution");

    static constexpr int E = (N == 64) ? 4 : OCEAN_FFT_E;
    static constexpr int T = N / E;
    static constexpr int NP = plan_passes(N, E);
    static constexpr int RL = N / ipow(E, NP - 1);
    static constexpr int M = E / RL;
    static constexpr int NS_LAST = N / RL;      // product of the radices before the last pass
    // LDS line length in complex elements for index padding i + (i >> PS) (no padding when swizzled instead)
    template<int PS> static constexpr int line() { return QuadFFT<N>::ENABLED ? N : N + (N >> PS); }

    static_assert(NP >= 2 && NP <= 6, "bad plan");
    static_assert(RL >= 2 && RL <= E && E % RL == 0, "bad plan");
    static_assert(ipow(E, NP - 1) * RL == N, "bad plan");
  };

  // LDS index padding: one extra element every 2^PS, breaks the power-of-two strides of the exchange stores.
  // PS = 4 is the compact choice (row pass: three workgroups per CU fit); PS = 3 has fewer bank conflicts with
  // the column pass's (column, row) lane order and is used there.
  template<int PS> OC_HD constexpr int padidx(int i) { return i + (i >> PS); }

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

      // t[1][c] *= exp(2 pi i c / 8)
      cf t1 = { (v[3].x - v[3].y) * h, (v[3].x + v[3].y) * h };
      cf t3 = { (-v[7].x - v[7].y) * h, (v[7].x - v[7].y) * h };

      cf o[8];
      o[0] = v[0] + v[1];  o[4] = v[0] - v[1];
      o[1] = v[2] + t1;    o[5] = v[2] - t1;
      o[2] = add_muli(v[4], v[5]);    o[6] = sub_muli(v[4], v[5]);
      o[3] = v[6] + t3;    o[7] = v[6] - t3;

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

  //|---------------------- one line ------------------------------------------
  // tw[k] = exp(+2 pi i k / N), k < N   (built in double on the host: ocean_capi)
  // `line` points at this line's padded LDS region (Plan::LINE elements)

  // per-thread twiddles kept in registers, each the first power of its pass; higher powers come from
  // twiddle_powers.  last[m] = exp(2 pi i j / N) for the last pass's tasks j = t + T m (Ns = N / RL);
  // mid[k] = exp(2 pi i (t % Ns) / (Ns E)) for middle pass k + 2 (Ns = E^(k+2)).
  template<int N>
  struct LineTwiddles
  {
    static constexpr int NMIDREG = (Plan<N>::NP > 3) ? Plan<N>::NP - 3 : 1;

    cf mid[NMIDREG];
    cf last[Plan<N>::M];
  };

  template<int N, int PS = 4>
  struct LineFFT
  {
    typedef Plan<N> P;

    static constexpr int LINE = P::template line<PS>();

    static constexpr int E = P::E;
    static constexpr int T = P::T;
    static constexpr int RL = P::RL;
    static constexpr int M = P::M;

    // per-thread twiddles kept in registers, each the first power of its pass; higher powers come from
    // twiddle_powers.  last[m] = exp(2 pi i j / N) for the last pass's tasks j = t + T m (Ns = N / RL);
    // mid[k] = exp(2 pi i (t % Ns) / (Ns E)) for middle pass k + 2 (Ns = E^(k+2)).
    static constexpr int NMIDREG = LineTwiddles<N>::NMIDREG;

    typedef LineTwiddles<N> Twiddles;

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
    static constexpr int MIDTAB = (P::NP >= 3 && !QuadFFT<N>::ENABLED) ? E * E : 0;

    static OC_HD cf midtab_entry(cf const *tw, int i)
    {
      int a = i % E, r = i / E;

      return tw[(a * r * (N / (E * E))) % N];
    }

    // pass 0: Ns = 1, radix E, task j = t.  v[s] = x[t + T s] -> line[t E + q]
    static OC_HD void pass0(cf (&v)[E], int t, cf *line)
    {
      Radix<E>::run(v);

      OC_UNROLL
      for(int q = 0; q < E; ++q)
        line[padidx<PS>(t * E + q)] = v[q];
    }

    // middle pass PASS (1 <= PASS <= NP-2): Ns = E^PASS, radix E, task j = t.  load + twiddle + butterfly
    template<int PASS>
    static OC_HD void mid_load(cf (&v)[E], int t, cf const *line, cf const *midtab, Twiddles const &w)
    {
      OC_UNROLL
      for(int r = 0; r < E; ++r)
        v[r] = line[padidx<PS>(t + T * r)];

      if (PASS == 1)
      {
        OC_UNROLL
        for(int r = 1; r < E; ++r)
          v[r] = cmul(v[r], midtab[r * E + (t % E)]);
      }
      else
      {
        constexpr int MI = (PASS >= 2 && PASS - 2 < NMIDREG) ? PASS - 2 : 0;

        cf p[E];
        twiddle_powers<E>(w.mid[MI], p);

        OC_UNROLL
        for(int r = 1; r < E; ++r)
          v[r] = cmul(v[r], p[r]);
      }

      Radix<E>::run(v);
    }

    template<int PASS>
    static OC_HD void mid_store(cf const (&v)[E], int t, cf *line)
    {
      constexpr int Ns = ipow(E, PASS);

      int base = (t / Ns) * Ns * E + (t % Ns);

      OC_UNROLL
      for(int q = 0; q < E; ++q)
        line[padidx<PS>(base + q * Ns)] = v[q];
    }

    // last pass: Ns = N/RL, radix RL, tasks j = t + T m.  result slot m + q M holds X[t + T (m + q M)]
    static OC_HD void last(cf (&v)[E], int t, cf const *line, Twiddles const &w)
    {
      OC_UNROLL
      for(int m = 0; m < M; ++m)
      {
        int j = t + T * m;

        cf u[RL];
        OC_UNROLL
        for(int r = 0; r < RL; ++r)
          u[r] = line[padidx<PS>(j + r * (N / RL))];

        cf p[RL];
        twiddle_powers<RL>(w.last[m], p);

        OC_UNROLL
        for(int r = 1; r < RL; ++r)
          u[r] = cmul(u[r], p[r]);

        Radix<RL>::run(u);

        OC_UNROLL
        for(int q = 0; q < RL; ++q)
          v[m + q * M] = u[q];
      }
    }
  };

  //|---------------------- N = 1024 as 32 x 32 with quad butterflies --------
  // Two radix-32 passes, ONE exchange through LDS (two barriers) instead of three exchanges (six barriers):
  // a radix-32 butterfly is done by the four lanes of a quad, each holding 8 of the 32 points (point r = a + 4b in
  // lane a, register b): radix-8 in registers over b, a twiddle exp(2 pi i q1 a / 32), then a radix-4 across the
  // four lanes with two DPP quad_perm exchanges (no LDS, no barrier).  Lane L ends with outputs q = q1 + 8 br(L)
  // in register q1, br = 2-bit reversal.
  //
  // Line ownership (T = 128 threads per line, thread t = 4 j + a):
  //   before: slot b holds element  j + 32 a + 128 b          (quad_elem_in)
  //   after:  slot q1 holds element j + 32 q1 + 256 br(a)     (quad_elem_out)

  OC_HD constexpr int bitrev2(int a) { return ((a & 1) << 1) | ((a >> 1) & 1); }

  OC_HD constexpr int quad_elem_in(int t, int s) { return (t >> 2) + 32 * (t & 3) + 128 * s; }
  OC_HD constexpr int quad_elem_out(int t, int s) { return (t >> 2) + 32 * s + 256 * bitrev2(t & 3); }

  // element a thread holds in slot s before / after a line transform
  template<int N> OC_HD constexpr int elem_in(int t, int s) { return QuadFFT<N>::ENABLED ? quad_elem_in(t, s) : t + Plan<N>::T * s; }
  template<int N> OC_HD constexpr int elem_out(int t, int s) { return QuadFFT<N>::ENABLED ? quad_elem_out(t, s) : t + Plan<N>::T * s; }

  // LDS position of line index i (10 bits) for the single exchange: an XOR swizzle under which both the pass-0
  // stores (index 32 j + 8 k + q1 over a 16-lane group: 4 j x 4 k) and the pass-1 loads (index j + 32 a + 128 b over
  // a 32-lane group: 8 j x 4 a) hit distinct banks.  P0 = b0^b3, P1 = b1^b4, P2 = b2^b6, P3 = b5, P4 = b6,
  // P5 = b3, P6 = b4, P7.. = b7..  (a bijection: no padding, a line is exactly 1024 values)
  OC_HD int quad_swizzle(int i)
  {
    int b3 = (i >> 3) & 1, b4 = (i >> 4) & 1, b5 = (i >> 5) & 1, b6 = (i >> 6) & 1;

    return ((i & 7) ^ (b3 | (b4 << 1) | (b6 << 2))) | (b5 << 3) | (b6 << 4) | (b3 << 5) | (b4 << 6) | (i & ~127);
  }

  struct QuadTwiddles
  {
    cf c32[8];     // exp(2 pi i q1 a / 32), a = lane in quad              (inside a radix-32 butterfly)
    cf pass1[8];   // exp(2 pi i j (a + 4 b) / 1024), j = t / 4            (between the two passes)
  };

  // tw[k] = exp(2 pi i k / 1024)
  OC_HD void quad_load_twiddles(cf const *tw, int t, QuadTwiddles &w)
  {
    int a = t & 3, j = t >> 2;

    OC_UNROLL
    for(int q = 0; q < 8; ++q)
    {
      w.c32[q] = tw[(32 * q * a) & 1023];
      w.pass1[q] = tw[(j * (a + 4 * q)) & 1023];
    }
  }

  // the part of a radix-32 butterfly that stays inside a lane: radix-8 over b, then the lane's twiddles
  OC_HD void quad_radix32_local(cf (&u)[8], QuadTwiddles const &w)
  {
    Radix<8>::run(u);

    OC_UNROLL
    for(int q = 1; q < 8; ++q)
      u[q] = cmul(u[q], w.c32[q]);
  }

  // one radix-2 step across lanes: own*sign + partner (sign = +1 in the lower lane of the pair, -1 in the upper)
  OC_HD cf quad_pair(cf own, cf partner, float sign) { return cf{ fmaf_(own.x, sign, partner.x), fmaf_(own.y, sign, partner.y) }; }

#if defined(__HIPCC__)
  // value of the lane whose quad index differs in bit 1 (XOR2: quad_perm [2,3,0,1]) or bit 0 (XOR1: [1,0,3,2])
  template<int CTRL>
  __device__ __forceinline__ float quad_xchg(float v)
  {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
#else
    return v;   // host pass of hipcc only parses this
#endif
  }

  // the cross-lane radix-4 over the quad (inverse transform: w4 = +i); lane L ends with output q2 = br(L)
  __device__ __forceinline__ void quad_radix4_lanes(cf (&u)[8], int a)
  {
    float const s1 = (a & 2) ? -1.0f : 1.0f;
    float const s2 = (a & 1) ? -1.0f : 1.0f;
    bool const rot = (a == 3);

    OC_UNROLL
    for(int q = 0; q < 8; ++q)
    {
      cf p = cf{ quad_xchg<0x4E>(u[q].x), quad_xchg<0x4E>(u[q].y) };
      cf e = quad_pair(u[q], p, s1);                 // lanes 0,1: Y_a + Y_(a+2); lanes 2,3: Y_(a-2) - Y_a

      e = rot ? cf{ -e.y, e.x } : e;                 // lane 3 holds Y1 - Y3: times i

      cf r = cf{ quad_xchg<0xB1>(e.x), quad_xchg<0xB1>(e.y) };
      u[q] = quad_pair(e, r, s2);                    // even lane: sum, odd lane: difference
    }
  }
#endif
}
