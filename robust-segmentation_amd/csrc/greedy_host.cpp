// K9 (host): greedy per-image attack selection that minimises the dataset mIoU.
//
// Replaces the pure-Python triple loop of evalSEA.worst_case_miou (tools/worse_only.py:279-334),
// which costs minutes to hours at ADE20K scale (N=2000, C=151) because every candidate builds
// torch tensors from Python lists.  The algorithm is inherently sequential (each accept changes the
// running totals), so it stays on the host; it is O(rounds * N * A * C) double operations.
//
// Bit-for-bit contract with the reference:
//   * table differences in float32, running totals rounded to float32 whenever the reference
//     rebuilds a tensor from its Python list, quotients in float64,
//   * candidate mIoU = mean_j (int_j + d_int_j) / (union_j + d_union_j + 1e-8) over classes whose OLD
//     running union is non-zero; accepted lists are the compacted ones (classes never seen are
//     dropped and later zips truncate - reference quirk D11, reproduced),
//   * the acceptance threshold is refreshed only after all attacks of an image were tried (D12),
//   * the mean is statistics.mean: the exactly rounded quotient of the exact rational sum,
//   * image order per round = CPython random.shuffle on the caller's Mersenne-Twister state.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/sea_hip.h"

namespace {

// ---- CPython-compatible MT19937 -----------------------------------------------------------------
struct MT {
  uint32_t* mt;  // 624 words + index at [624]
  uint32_t next() {
    uint32_t& idx = mt[624];
    if (idx >= 624) {
      for (int k = 0; k < 624; ++k) {
        uint32_t yv = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (yv >> 1) ^ ((yv & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t yv = mt[idx++];
    yv ^= (yv >> 11);
    yv ^= (yv << 7) & 0x9d2c5680u;
    yv ^= (yv << 15) & 0xefc60000u;
    yv ^= (yv >> 18);
    return yv;
  }
  // Random._randbelow_with_getrandbits for n < 2^32
  uint32_t randbelow(uint32_t n) {
    int k = 0;
    for (uint32_t t = n; t; t >>= 1) ++k;  // n.bit_length()
    uint32_t r = next() >> (32 - k);
    while (r >= n) r = next() >> (32 - k);
    return r;
  }
  void shuffle(std::vector<int32_t>& x) {
    for (size_t i = x.size() - 1; i >= 1; --i) {
      uint32_t j = randbelow((uint32_t)(i + 1));
      std::swap(x[i], x[j]);
    }
  }
};

// ---- exactly rounded mean of doubles (statistics.mean) --------------------------------------------
// Fixed-point super-accumulator: value = acc * 2^-1074, 2240 bits.
struct ExactSum {
  static constexpr int L = 35;
  uint64_t pos[L], neg[L];
  ExactSum() {
    std::memset(pos, 0, sizeof(pos));
    std::memset(neg, 0, sizeof(neg));
  }
  static void add_to(uint64_t* a, uint64_t mant, int shift) {
    int limb = shift / 64, off = shift % 64;
    unsigned __int128 v = (unsigned __int128)mant << off;
    uint64_t lo = (uint64_t)v, hi = (uint64_t)(v >> 64);
    unsigned __int128 c = (unsigned __int128)a[limb] + lo;
    a[limb] = (uint64_t)c;
    uint64_t carry = (uint64_t)(c >> 64);
    c = (unsigned __int128)a[limb + 1] + hi + carry;
    a[limb + 1] = (uint64_t)c;
    carry = (uint64_t)(c >> 64);
    for (int i = limb + 2; carry && i < L; ++i) {
      c = (unsigned __int128)a[i] + carry;
      a[i] = (uint64_t)c;
      carry = (uint64_t)(c >> 64);
    }
  }
  void add(double x) {
    if (x == 0.0) return;
    int e;
    double m = std::frexp(std::fabs(x), &e);           // |x| = m * 2^e, m in [0.5,1)
    uint64_t mant = (uint64_t)std::ldexp(m, 53);       // 53-bit integer
    int shift = e - 53 + 1074;                          // |x| = mant * 2^(e-53)
    while (shift < 0) {                                 // subnormal tail: exact, low bits are zero
      mant >>= 1;
      ++shift;
    }
    add_to(x > 0 ? pos : neg, mant, shift);
  }
  // round_to_nearest_even((pos - neg) / n)
  double mean(uint32_t n) const {
    uint64_t d[L];
    bool negative = false;
    // d = pos - neg (or neg - pos)
    int cmp = 0;
    for (int i = L - 1; i >= 0; --i)
      if (pos[i] != neg[i]) {
        cmp = pos[i] > neg[i] ? 1 : -1;
        break;
      }
    if (cmp == 0) return 0.0;
    const uint64_t* a = cmp > 0 ? pos : neg;
    const uint64_t* b = cmp > 0 ? neg : pos;
    negative = cmp < 0;
    uint64_t borrow = 0;
    for (int i = 0; i < L; ++i) {
      unsigned __int128 t = (unsigned __int128)a[i] - b[i] - borrow;
      d[i] = (uint64_t)t;
      borrow = (uint64_t)((t >> 64) & 1);
    }
    // q = d / n, rem
    uint64_t q[L];
    unsigned __int128 rem = 0;
    for (int i = L - 1; i >= 0; --i) {
      unsigned __int128 cur = (rem << 64) | d[i];
      q[i] = (uint64_t)(cur / n);
      rem = cur % n;
    }
    // top bit of q
    int top = -1;
    for (int i = L - 1; i >= 0 && top < 0; --i)
      if (q[i]) top = i * 64 + 63 - __builtin_clzll(q[i]);
    if (top < 0) {
      // |value| < 2^-1074: round to 0 or the smallest subnormal
      double r = ((unsigned __int128)rem * 2 > n) ? std::ldexp(1.0, -1074) : 0.0;
      return negative ? -r : r;
    }
    auto bit = [&](int p) -> int { return p < 0 ? 0 : (int)((q[p / 64] >> (p % 64)) & 1); };
    uint64_t mant = 0;
    int lowest = top - 52;  // position of the mantissa LSB
    for (int p = top; p >= lowest && p >= 0; --p) mant = (mant << 1) | (uint64_t)bit(p);
    if (lowest < 0) {
      // fewer than 53 significant bits available: value is q exactly plus remainder/n
      // rounding happens at bit 0 (2^-1074 grid is coarser than needed only for subnormals)
      mant = 0;
      for (int p = top; p >= 0; --p) mant = (mant << 1) | (uint64_t)bit(p);
      // fraction below bit 0 is rem/n
      unsigned __int128 twice = (unsigned __int128)rem * 2;
      if (twice > n || (twice == n && (mant & 1))) ++mant;
      double r = std::ldexp((double)mant, -1074);
      return negative ? -r : r;
    }
    int guard = bit(lowest - 1);
    bool sticky = rem != 0;
    for (int p = lowest - 2; p >= 0 && !sticky; --p) sticky = bit(p) != 0;
    if (guard && (sticky || (mant & 1))) ++mant;  // may carry to 2^53: ldexp handles it exactly
    double r = std::ldexp((double)mant, lowest - 1074);
    return negative ? -r : r;
  }
};

double exact_mean(const std::vector<double>& v) {
  ExactSum s;
  for (double x : v) s.add(x);
  return s.mean((uint32_t)v.size());
}

inline double f32(double x) { return (double)(float)x; }

// _compute_miou (worse_only.py:69-76) on float32-rounded lists
bool miou_plain(const std::vector<double>& ri, const std::vector<double>& ru, double* out) {
  std::vector<double> vals;
  size_t n = ri.size() < ru.size() ? ri.size() : ru.size();
  vals.reserve(n);
  for (size_t j = 0; j < n; ++j) {
    double b = f32(ru[j]);
    if (b == 0.0) continue;
    vals.push_back(f32(ri[j]) / b);
  }
  if (vals.empty()) return false;
  *out = exact_mean(vals);
  return true;
}

}  // namespace

extern "C" int sea_worst_miou_greedy(const float* ints, const float* unions, int A, int N, int C,
                                     uint32_t* mt_state, int n_rounds, double* miou, int32_t* selected,
                                     int32_t* rounds_run) {
  if (!ints || !unions || !mt_state || !miou || !selected || A <= 0 || N <= 0 || C <= 0 || n_rounds < 0) return 1;
  MT rng{mt_state};
  auto T = [&](const float* t, int a, int n) { return t + ((size_t)a * N + n) * C; };

  // running totals from attack 0, accumulated image by image in float32 (worse_only.py:241-250)
  std::vector<float> acc_i(C, 0.f), acc_u(C, 0.f);
  for (int n = 0; n < N; ++n) {
    const float* ti = T(ints, 0, n);
    const float* tu = T(unions, 0, n);
    for (int c = 0; c < C; ++c) {
      acc_i[c] += ti[c];
      acc_u[c] += tu[c];
    }
  }
  std::vector<double> run_i(acc_i.begin(), acc_i.end()), run_u(acc_u.begin(), acc_u.end());
  double final_miou;
  if (!miou_plain(run_i, run_u, &final_miou)) return 1;
  for (int n = 0; n < N; ++n) selected[n] = 0;

  double prev_best = 10.0;
  int rounds = 0;
  std::vector<int32_t> order(N);
  std::vector<double> new_i, new_u, vals;
  for (int r = 0; r < n_rounds; ++r) {
    ++rounds;
    for (int n = 0; n < N; ++n) order[n] = n;
    if (N > 1) rng.shuffle(order);
    for (int oi = 0; oi < N; ++oi) {
      const int idx = order[oi];
      for (int a = 0; a < A; ++a) {
        const float* ia = T(ints, a, idx);
        const float* is = T(ints, selected[idx], idx);
        const float* ua = T(unions, a, idx);
        const float* us = T(unions, selected[idx], idx);
        const size_t L = run_i.size() < (size_t)C ? run_i.size() : (size_t)C;
        new_i.clear();
        new_u.clear();
        vals.clear();
        for (size_t j = 0; j < L; ++j) {
          const double b = f32(run_u[j]);
          if (b == 0.0) continue;
          const double di = (double)(float)(ia[j] - is[j]);
          const double du = (double)(float)(ua[j] - us[j]);
          const double ni = f32(run_i[j]) + di;
          const double nu = b + du;
          new_i.push_back(ni);
          new_u.push_back(nu);
          vals.push_back(ni / (nu + 1e-8));
        }
        if (vals.empty()) return 1;
        const double est = exact_mean(vals);
        if (est < final_miou) {
          selected[idx] = a;
          run_i = new_i;
          run_u = new_u;
        }
      }
      if (!miou_plain(run_i, run_u, &final_miou)) return 1;
    }
    if (prev_best - final_miou <= 1e-6) break;
    prev_best = final_miou;
  }
  *miou = final_miou;
  if (rounds_run) *rounds_run = rounds;
  return 0;
}
