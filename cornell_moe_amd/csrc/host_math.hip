// cornell_moe_amd/csrc/host_math.hip -- see host_math.hpp.  Host code; compiled as HIP only to share the covariance
// scalar functions (device_cov.hpp) with the kernels so host and device evaluate identical formulas.
#include "host_math.hpp"

#include <cmath>
#include <cstring>

#include "device_cov.hpp"

namespace moe {

namespace {
constexpr int HD = kMaxDimPadded;

inline Radial host_radial(const CovParams& cp, const double* p1, const double* p2, double (&diff)[HD]) {
  double r2 = 0.0;
  for (int k = 0; k < HD; ++k) {
    diff[k] = (k < cp.dim) ? (p1[k] - p2[k]) : 0.0;
    r2 = std::fma(diff[k] * diff[k], cp.inv_l2[k], r2);
  }
  return radial_scalars(cp.type, cp.alpha, r2);
}
}  // namespace

void host_cov(const CovParams& cp, const double* p1, const DerivList& d1, const double* p2, const DerivList& d2, double* out) {
  double diff[HD];
  const Radial rd = host_radial(cp, p1, p2, diff);
  for (int a = 0; a < 1 + d1.g; ++a)
    for (int b = 0; b < 1 + d2.g; ++b) out[a + b * (1 + d1.g)] = cov_entry<HD>(cp, rd, diff, a, b, d1, d2);
}

void host_grad_cov(const CovParams& cp, const double* p1, const DerivList& d1, const double* p2, const DerivList& d2,
                   double* out) {
  double diff[HD];
  const Radial rd = host_radial(cp, p1, p2, diff);
  for (int a = 0; a < 1 + d1.g; ++a)
    for (int b = 0; b < 1 + d2.g; ++b)
      for (int dd = 0; dd < cp.dim; ++dd)
        out[dd + a * cp.dim + b * cp.dim * (1 + d1.g)] = grad_cov_entry<HD>(cp, rd, diff, a, b, dd, d1, d2);
}

void host_mean(const StateHost& s, double* mu) {
  const StateLayout& L = s.lay;
  for (int j = 0; j < L.u; ++j)
    for (int b = 0; b < 1 + L.gt; ++b) mu[j * (1 + L.gt) + b] = ((b == 0) ? s.mean : 0.0) + s.ek[L.col_kstar(j, b)];
}

void host_grad_mean(const StateHost& s, double* out) {
  const StateLayout& L = s.lay;
  for (int i = 0; i < L.nd; ++i)
    for (int a = 0; a < 1 + L.gt; ++a)
      for (int dd = 0; dd < L.d; ++dd) out[dd + (size_t)(i * (1 + L.gt) + a) * L.d] = s.ek[L.col_grad(i, a, dd)];
}

void host_variance(const StateHost& s, double* var) {
  const StateLayout& L = s.lay;
  const int m = L.m, gt = L.gt;
  std::vector<double> blk((size_t)(1 + gt) * (1 + gt));
  for (int j = 0; j < L.u; ++j) {
    for (int i = 0; i < L.u; ++i) {
      host_cov(s.cp, &s.U[(size_t)i * L.d], s.dt, &s.U[(size_t)j * L.d], s.dt, blk.data());
      for (int a = 0; a < 1 + gt; ++a)
        for (int b = 0; b < 1 + gt; ++b) {
          const int row = i * (1 + gt) + a, col = j * (1 + gt) + b;
          var[row + (size_t)col * m] = blk[a + b * (1 + gt)] - s.G(L.col_kstar(i, a), L.col_kstar(j, b));
        }
    }
  }
}

void host_grad_variance_per_point(const StateHost& s, int p, double* gv) {
  const StateLayout& L = s.lay;
  const int d = L.d, gt = L.gt, m = L.m, u = L.u;
  std::memset(gv, 0, sizeof(double) * (size_t)d * m * m);
  // column block p:  -(d K*_p)^T K^-1 K*_j      (gpp_math.cpp:1277-1292)
  for (int a = 0; a < 1 + gt; ++a) {
    const int col = p * (1 + gt) + a;
    for (int row = 0; row < m; ++row)
      for (int dd = 0; dd < d; ++dd) gv[dd + (size_t)row * d + (size_t)col * d * m] = -s.G(L.col_grad(p, a, dd), row);
  }
  // (p,p) block: both factors depend on Xs_p       (gpp_math.cpp:1293-1302)
  for (int a = 0; a < 1 + gt; ++a)
    for (int b = a; b < 1 + gt; ++b)
      for (int dd = 0; dd < d; ++dd) {
        const size_t row = (size_t)p * (1 + gt) + a, col = (size_t)p * (1 + gt) + b;
        gv[dd + row * d + col * d * m] += gv[dd + col * d + row * d * m];
        gv[dd + col * d + row * d * m] = gv[dd + row * d + col * d * m];
      }
  // + d Kss / d Xs_p                                (gpp_math.cpp:1303-1327)
  std::vector<double> tmp((size_t)d * (1 + gt) * (1 + gt));
  for (int j = 0; j < u; ++j) {
    host_grad_cov(s.cp, &s.U[(size_t)p * d], s.dt, &s.U[(size_t)j * d], s.dt, tmp.data());
    for (int a = 0; a < 1 + gt; ++a)
      for (int b = 0; b < 1 + gt; ++b) {
        const size_t row = (size_t)j * (1 + gt) + a, col = (size_t)p * (1 + gt) + b;
        for (int dd = 0; dd < d; ++dd) {
          if (j == p)
            gv[dd + row * d + col * d * m] += tmp[dd + b * d + a * d * (1 + gt)] + tmp[dd + a * d + b * d * (1 + gt)];
          else
            gv[dd + row * d + col * d * m] += tmp[dd + b * d + a * d * (1 + gt)];
        }
      }
  }
  // mirror block column p into block row p          (gpp_math.cpp:1328-1343)
  for (int i = 0; i < 1 + gt; ++i) {
    const size_t row = (size_t)p * (1 + gt) + i;
    for (int j = 0; j < u; ++j)
      for (int b = 0; b < 1 + gt; ++b) {
        const size_t col = (size_t)j * (1 + gt) + b;
        if (j != p)
          for (int dd = 0; dd < d; ++dd) gv[dd + d * row + (size_t)d * m * col] = gv[dd + d * col + (size_t)d * m * row];
      }
  }
}

int host_cholesky(int n, double* a) {
  for (int k = 0; k < n; ++k) {
    double* col = a + (size_t)k * n;
    if (col[k] > 1.0e-16) {
      const double akk = std::sqrt(col[k]);
      col[k] = akk;
      for (int j = k + 1; j < n; ++j) col[j] /= akk;
      for (int j = k + 1; j < n; ++j) {
        double* cj = a + (size_t)j * n;
        const double ljk = col[j];
        for (int i = j; i < n; ++i) cj[i] = cj[i] - col[i] * ljk;
      }
    } else {
      return k + 1;
    }
  }
  return 0;
}

void host_grad_cholesky_per_point(const StateHost& s, int p, const double* chol, double* gc) {
  const int d = s.lay.d, m = s.lay.m;
  const double kMinimumStdDev = 2.220446049250313e-16;  // gpp_math.hpp:291
  host_grad_variance_per_point(s, p, gc);
  for (int i = 0; i < m; ++i) {
    double* col = gc + (size_t)i * m * d;
    for (int j = (i + 1) * d; j < d * m; ++j) col[j] = 0.0;
  }
#define CH(i, j) chol[(size_t)(j)*m + (i)]
#define GC(dd, i, j) gc[(size_t)(j)*m * d + (size_t)(i)*d + (dd)]
  for (int k = 0; k < m; ++k) {
    const double Lkk = CH(k, k);
    if (Lkk > kMinimumStdDev) {
      for (int dd = 0; dd < d; ++dd) GC(dd, k, k) = 0.5 * GC(dd, k, k) / Lkk;
      for (int j = k + 1; j < m; ++j)
        for (int dd = 0; dd < d; ++dd) GC(dd, k, j) = (GC(dd, k, j) - CH(j, k) * GC(dd, k, k)) / Lkk;
      for (int j = k + 1; j < m; ++j)
        for (int i = j; i < m; ++i)
          for (int dd = 0; dd < d; ++dd) GC(dd, j, i) = GC(dd, j, i) - GC(dd, k, i) * CH(j, k) - CH(i, k) * GC(dd, k, j);
    }
  }
#undef CH
#undef GC
}

void host_tri_solve(const double* A, char trans, int n, double* x) {
  if (trans == 'N') {
    for (int j = 0; j < n; ++j) {
      if (x[j] != 0.0) {
        x[j] /= A[j + (size_t)j * n];
        const double t = x[j];
        for (int i = j + 1; i < n; ++i) x[i] -= t * A[i + (size_t)j * n];
      }
    }
  } else {
    for (int j = n - 1; j >= 0; --j) {
      double t = x[j];
      for (int i = n - 1; i >= j + 1; --i) t -= A[i + (size_t)j * n] * x[i];
      x[j] = t / A[j + (size_t)j * n];
    }
  }
}

}  // namespace moe
