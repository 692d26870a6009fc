// CPU emulation of the thread-parallel line FFT in datum_amd/csrc/ocean_fft_core.h:
// the T threads of a line are run one after another, phase by phase (a phase boundary is where
// the kernels put a barrier), with an ordinary array standing in for the LDS line.
// Test harness only: exercises the PRODUCT's index arithmetic and butterflies without a GPU.
#include <cmath>
#include <vector>
#include "../../datum_amd/csrc/ocean_fft_core.h"

using namespace ocean;

// W = lines interleaved element by element in the LDS array (the column pass's columns); the line emulated is column W - 1
template<int N, int E = default_radix(N), int W = 1>
static void run_line(float const *in, float *out)
{
  typedef LineFFT<N, W, E> L;
  typedef Plan<N, E> P;

  std::vector<cf> tw(N);
  for(int k = 0; k < N; ++k)
  {
    double a = 2.0 * M_PI * k / N;
    tw[k] = cf{ (float)std::cos(a), (float)std::sin(a) };
  }

  std::vector<cf> linebuf((size_t)L::LINE * W, cf{ 1e30f, -1e30f });
  cf *const linep = linebuf.data() + (W - 1);
  std::vector<cf> regs(N);   // [t][s]
  std::vector<typename L::Twiddles> w(P::T);
  std::vector<cf> midtab(L::MIDTAB + 1);
  for(int i = 0; i < L::MIDTAB; ++i)
    midtab[i] = L::midtab_entry(tw.data(), i);

  for(int t = 0; t < P::T; ++t)
  {
    L::load_twiddles(tw.data(), t, w[t]);
    for(int s = 0; s < P::E; ++s)
      regs[t*P::E + s] = cf{ in[2*(t + P::T*s)], in[2*(t + P::T*s)+1] };
  }

  auto R = [&](int t) -> cf (&)[P::E] { return *reinterpret_cast<cf (*)[P::E]>(&regs[t*P::E]); };

  for(int t = 0; t < P::T; ++t) L::pass0(R(t), t, linep);
  if (P::NP >= 3)
  {
    for(int t = 0; t < P::T; ++t) L::template mid_load<1>(R(t), t, linep, midtab.data(), w[t]);
    for(int t = 0; t < P::T; ++t) L::template mid_store<1>(R(t), t, linep);
  }
  if (P::NP >= 4)
  {
    for(int t = 0; t < P::T; ++t) L::template mid_load<2>(R(t), t, linep, midtab.data(), w[t]);
    for(int t = 0; t < P::T; ++t) L::template mid_store<2>(R(t), t, linep);
  }
  if (P::NP >= 5)
  {
    for(int t = 0; t < P::T; ++t) L::template mid_load<3>(R(t), t, linep, midtab.data(), w[t]);
    for(int t = 0; t < P::T; ++t) L::template mid_store<3>(R(t), t, linep);
  }
  if (P::NP >= 6)
  {
    for(int t = 0; t < P::T; ++t) L::template mid_load<4>(R(t), t, linep, midtab.data(), w[t]);
    for(int t = 0; t < P::T; ++t) L::template mid_store<4>(R(t), t, linep);
  }
  for(int t = 0; t < P::T; ++t) L::last(R(t), t, linep, w[t]);

  for(int t = 0; t < P::T; ++t)
    for(int s = 0; s < P::E; ++s)
    {
      out[2*(t + P::T*s)] = regs[t*P::E + s].x;
      out[2*(t + P::T*s)+1] = regs[t*P::E + s].y;
    }
}

extern "C" int emul_line_ifft(int N, float const *in, float *out)
{
  switch(N)
  {
    case 64: run_line<64>(in, out); break;
    case 128: run_line<128>(in, out); break;
    case 256: run_line<256>(in, out); break;
    case 512: run_line<512>(in, out); break;
    case 1024: run_line<1024>(in, out); break;
    case 2048: run_line<2048>(in, out); break;
    case 4096: run_line<4096>(in, out); break;
    default: return -1;
  }
  return 0;
}

// the same with 16 points per thread (the column pass of the largest grids: ColCfg::E)
extern "C" int emul_line_ifft16(int N, float const *in, float *out)
{
  switch(N)
  {
    case 256: run_line<256, 16>(in, out); break;
    case 512: run_line<512, 16>(in, out); break;
    case 1024: run_line<1024, 16>(in, out); break;
    case 2048: run_line<2048, 16>(in, out); break;
    case 4096: run_line<4096, 16>(in, out); break;
    default: return -1;
  }
  return 0;
}

// the same over the interleaved layouts of the column pass (W columns element by element), radix 8 or 16
template<int E, int W>
static int run_w(int N, float const *in, float *out)
{
  switch(N)
  {
    case 256: run_line<256, E, W>(in, out); break;
    case 512: run_line<512, E, W>(in, out); break;
    case 1024: run_line<1024, E, W>(in, out); break;
    case 2048: run_line<2048, E, W>(in, out); break;
    case 4096: run_line<4096, E, W>(in, out); break;
    default: return -1;
  }
  return 0;
}

extern "C" int emul_line_ifft_w(int N, int E, int W, float const *in, float *out)
{
  if (E == 8 && W == 2) return run_w<8, 2>(N, in, out);
  if (E == 8 && W == 4) return run_w<8, 4>(N, in, out);
  if (E == 8 && W == 8) return run_w<8, 8>(N, in, out);
  if (E == 16 && W == 2) return run_w<16, 2>(N, in, out);
  if (E == 16 && W == 4) return run_w<16, 4>(N, in, out);
  return -1;
}
