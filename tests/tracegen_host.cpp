// TEST-ONLY host build of the chip-table row writers (vectorx_amd/csrc/tracegen_core.h + tracegen_prep.h): the same functions the
// device kernels of tracegen.hip.h call, compiled with g++ so that tests/test_tracegen.py can compare them cell by cell with the
// numpy generators WITHOUT a GPU.  Not part of the product: libvxprover.so has no host trace generator (nothing under vectorx_amd/
// loads this file's library).   g++ -O2 -shared -fPIC -o tests/libtracegen_host.so tests/tracegen_host.cpp
#include "../vectorx_amd/csrc/tracegen_prep.h"
#include "../vectorx_amd/csrc/tracegen_eddsa.h"

template <class T>
static int sha2_host(int degree_bits, const uint8_t* msgs, const uint64_t* off, int nmsg, uint64_t* trace, uint64_t* pis, uint64_t* digest_words) {
  tg::Sha2Prep<T> prep;
  const int pr = tg::sha2_prepare<T>(degree_bits, msgs, off, nmsg, prep);
  if (pr) return pr;
  const size_t n = (size_t)1 << degree_bits;
  std::vector<tg::Sha2Expanded<T>> exp(prep.blocks.size());
  for (size_t b = 0; b < prep.blocks.size(); ++b) tg::sha2_expand<T>(prep.blocks[b], tg::Sha2Consts<T>::K(), exp[b]);
  uint64_t hist[8] = {0};
  for (size_t row = 0; row < n; ++row) {
    const size_t b = row / T::PERIOD;
    unsigned carries[T::NCARRY];
    tg::sha2_row<T>(prep.blocks[b], exp[b], b ? &exp[b - 1] : nullptr, (int)(row % T::PERIOD), row, tg::Sha2Consts<T>::K(),
                    [&](int col, uint64_t v) { trace[(size_t)col * n + row] = v; }, carries);
    if (row + 1 < n)
      for (int q = 0; q < T::NCARRY; ++q) hist[carries[q] & 7]++;
  }
  for (int i = 0; i < 8; ++i) trace[(size_t)T::MULT * n + i] = hist[i];
  memcpy(pis, prep.pis, sizeof prep.pis);
  for (size_t i = 0; i < prep.digests.size(); ++i) digest_words[i] = (uint64_t)prep.digests[i];
  return 0;
}

extern "C" {
int tgh_sha256(int degree_bits, const uint8_t* msgs, const uint64_t* off, int nmsg, uint64_t* trace, uint64_t* pis, uint64_t* digest_words) {
  return sha2_host<tg::Sha256T>(degree_bits, msgs, off, nmsg, trace, pis, digest_words);
}
int tgh_sha512(int degree_bits, const uint8_t* msgs, const uint64_t* off, int nmsg, uint64_t* trace, uint64_t* pis, uint64_t* digest_words) {
  return sha2_host<tg::Sha512T>(degree_bits, msgs, off, nmsg, trace, pis, digest_words);
}
int tgh_sha512_bus(int degree_bits, const uint8_t* msgs, const uint64_t* off, int nmsg, uint64_t* trace, uint64_t* pis, uint64_t* digest_words) {
  return sha2_host<tg::Sha512BusT>(degree_bits, msgs, off, nmsg, trace, pis, digest_words);
}
int tgh_blake2b(int degree_bits, const uint8_t* msgs, const uint64_t* off, int nmsg, uint64_t* trace, uint64_t* pis, uint64_t* digest_words) {
  tg::B2Prep prep;
  const int pr = tg::b2_prepare(degree_bits, msgs, off, nmsg, prep);
  if (pr) return pr;
  const size_t n = (size_t)1 << degree_bits;
  std::vector<tg::b2::Expanded> exp(prep.blocks.size());
  for (size_t b = 0; b < prep.blocks.size(); ++b) tg::b2::expand(prep.blocks[b], tg::B2_IV, tg::B2_SIGMA, exp[b]);
  std::vector<uint64_t> hist(65536, 0);
  const uint64_t zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t row = 0; row < n; ++row) {
    const size_t b = row / tg::b2::PERIOD;
    const tg::b2::Block& blk = prep.blocks[b];
    const bool count = row + 1 < n;
    tg::b2::row(blk, exp[b], b ? exp[b - 1].hn : zero8, blk.dsrc >= 0 ? exp[blk.dsrc].hn : zero8, (int)(row % tg::b2::PERIOD), row, tg::B2_IV,
                tg::B2_SIGMA, [&](int col, uint64_t v) { trace[(size_t)col * n + row] = v; },
                [&](unsigned a, unsigned bb) { if (count) hist[a * 256u + bb]++; });
  }
  for (int k = 0; k < tg::b2::NTAB; ++k)
    for (int i = 0; i < tg::b2::TAB_ROWS; ++i) trace[(size_t)tg::b2::tabcol(k, 19) * n + i] = hist[(size_t)tg::b2::TAB_ROWS * k + i];
  memcpy(pis, prep.pis, sizeof prep.pis);
  for (size_t i = 0; i < prep.digests.size(); ++i) digest_words[i] = prep.digests[i];
  return 0;
}
int tgh_eddsa(int degree_bits, int scalar_bits, int full, const uint64_t* sigs, int nsig, uint64_t* trace, uint64_t* results) {
  const tg::ed::Cols cl = tg::ed::cols(scalar_bits, full);
  const size_t n = (size_t)1 << degree_bits;
  if ((size_t)nsig > (n - 1) / cl.L) return 1;
  const int ninst = (int)((n + cl.L - 1) / cl.L);
  tg::ed::Sig filler;
  memset(&filler, 0, sizeof filler);
  for (int k = 0; k < 4; ++k) filler.ax[k] = tg::ed::BX[k], filler.ay[k] = tg::ed::BY[k];
  const int stride = full ? 24 : 16;
  std::vector<tg::ed::Sig> sg((size_t)(nsig ? nsig : 1));
  memset(sg.data(), 0, sg.size() * sizeof(tg::ed::Sig));
  for (int i = 0; i < nsig; ++i) {
    const uint64_t* w = sigs + (size_t)i * stride;
    memcpy(sg[i].ax, w, 32), memcpy(sg[i].ay, w + 4, 32), memcpy(sg[i].s, w + 8, 32), memcpy(sg[i].h, w + 12, 32);
    if (full) memcpy(sg[i].d, w + 16, 64);
  }
  std::vector<tg::ed::RowVals> vals((size_t)ninst * cl.L);
  uint64_t regs[tg::ed::NREG][4];
  for (int u = 0; u < ninst; ++u) {
    const int why = tg::ed::simulate_instance(cl, u < nsig ? sg[u] : filler, vals.data() + (size_t)u * cl.L, regs);
    if (why && u < nsig) return 10 + why;
  }
  tg::ed::RegSrc rsrc;
  memset(&rsrc, 0, sizeof rsrc);
  tg::ed::make_reg_src(cl, rsrc);
  std::vector<uint64_t> hist(65536, 0);
  for (size_t row = 0; row < n; ++row) {
    const bool count = row + 1 < n;
    tg::ed::row(cl, rsrc, vals.data(), sg.data(), nsig, filler, row, [&](int col, uint64_t v) { trace[(size_t)col * n + row] = v; },
                [&](unsigned limb) { if (count) hist[limb]++; });
  }
  for (int i = 0; i < 65536; ++i) trace[(size_t)cl.MULT * n + i] = hist[i];
  for (int u = 0; u < nsig; ++u)
    for (int w = 0; w < 2; ++w) memcpy(results + ((size_t)u * 2 + w) * 4, vals[(size_t)u * cl.L + cl.XROW + w].z, 32);
  return 0;
}
}
