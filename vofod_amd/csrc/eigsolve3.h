// Eigen-decomposition of the cluster covariance the way pcl::MomentOfInertiaEstimation gets it (product code, host and
// device).  vofod_nodelet.cpp:1654-1673 calls pcl::MomentOfInertiaEstimation, whose computeEigenVectors runs
// Eigen::EigenSolver<Matrix3f> - the GENERAL real solver (Householder Hessenberg reduction, Francis double-shift QR to the real
// Schur form, back substitution), in float.  For MAV-sized clusters (2-10 lattice points) the covariance very often has a
// repeated eigenvalue; inside such an eigen-space every solver picks its own basis, and the OBB centre / extents that
// classify_cluster gates on (:1689, :1696) and that the detection reports (:851) move with that choice (round 3: a symmetric
// Jacobi solver differed from this path by up to 0.13 m on 5.6 % of random lattice clusters).  So the product follows
// Eigen 3.3.7's operation sequence step by step, in float, every product and sum rounded separately (the library is built
// with -ffp-contract=off; the device's float division and square root are correctly rounded).
//
// Matrices are flat row-major float[9]; one reflector routine serves rows and columns through strides.
#pragma once
#include <cfloat>
#include <cmath>

#ifdef __HIPCC__
#define VE_HD __host__ __device__
#else
#define VE_HD
#endif

namespace ve
{

VE_HD inline float fabs_(float x) { return x < 0.0f ? -x : x; }
VE_HD inline float fmax_(float a, float b) { return a < b ? b : a; }

// Householder vector of x[0..n): x -> (beta, 0, ..), essential part e[0..n-1), factor tau (Eigen makeHouseholder)
VE_HD inline void reflector(const float* x, int n, float* e, float& tau, float& beta)
{
  float tail = 0.0f;
  for (int i = 1; i < n; i++)
    tail += x[i] * x[i];
  const float head = x[0];
  if (tail <= FLT_MIN)
  {
    tau = 0.0f;
    beta = head;
    for (int i = 0; i + 1 < n; i++)
      e[i] = 0.0f;
    return;
  }
  beta = sqrtf(head * head + tail);
  if (head >= 0.0f)
    beta = -beta;
  for (int i = 0; i + 1 < n; i++)
    e[i] = x[i + 1] / (head - beta);
  tau = (beta - head) / beta;
}

// Applies H = I - tau (1, e)(1, e)^T to `len` vectors of `n` elements each.  Element i of vector k sits at
// base[i * es + k * vs].  `left` selects Eigen's association of the update: (tau * e_i) * tmp for a reflection applied from
// the left (vectors = columns), (tau * tmp) * e_i from the right (vectors = rows).
VE_HD inline void reflect(float* base, int es, int vs, int n, int len, const float* e, float tau, bool left)
{
  if (n == 1)
  {
    for (int k = 0; k < len; k++)
      base[k * vs] *= 1.0f - tau;
    return;
  }
  if (tau == 0.0f)
    return;
  for (int k = 0; k < len; k++)
  {
    float* v = base + k * vs;
    float t = 0.0f;
    for (int i = 1; i < n; i++)
      t += e[i - 1] * v[i * es];
    t += v[0];
    v[0] -= tau * t;
    for (int i = 1; i < n; i++)
      v[i * es] -= left ? (tau * e[i - 1]) * t : (tau * t) * e[i - 1];
  }
}

// plane rotation of Eigen's JacobiRotation::makeGivens (real case)
VE_HD inline void givens(float p, float q, float& c, float& s)
{
  if (q == 0.0f)
  {
    c = p < 0.0f ? -1.0f : 1.0f;
    s = 0.0f;
    return;
  }
  if (p == 0.0f)
  {
    c = 0.0f;
    s = q < 0.0f ? 1.0f : -1.0f;
    return;
  }
  if (fabs_(p) > fabs_(q))
  {
    const float t = q / p;
    float u = sqrtf(1.0f + t * t);
    u = p < 0.0f ? -u : u;
    c = 1.0f / u;
    s = -t * c;
  }
  else
  {
    const float t = p / q;
    float u = sqrtf(1.0f + t * t);
    u = q < 0.0f ? -u : u;
    s = -1.0f / u;
    c = -t * s;
  }
}

// (x, y) <- (c x - s y, s x + c y) over `len` pairs
VE_HD inline void rotate(float* x, float* y, int stride, int len, float c, float s)
{
  for (int k = 0; k < len; k++)
  {
    const float a = x[k * stride], b = y[k * stride];
    x[k * stride] = c * a - s * b;
    y[k * stride] = s * a + c * b;
  }
}

// Eigenvalues (real parts, in the solver's order) and the real parts of the normalised eigenvectors (column j of vec) of a
// real 3 x 3 matrix, as Eigen::EigenSolver<Matrix3f>(cov).eigenvalues().real() / .eigenvectors().real() deliver them.
VE_HD inline void eigsolve3(const float cov[3][3], float val[3], float vec[3][3])
{
  const float eps = FLT_EPSILON;
  float t[9], u[9] = {1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f};
  float big = 0.0f;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      big = fmax_(big, fabs_(cov[i][j]));
  for (int i = 0; i < 9; i++)
    t[i] = 0.0f;
  if (!(big < FLT_MIN))
  {
    // Hessenberg form of cov / big: one reflector on (a10, a20), applied from both sides; Q is that reflector
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        t[3 * i + j] = cov[i][j] / big;
    {
      const float x[2] = {t[3], t[6]};
      float e[1], tau, beta;
      reflector(x, 2, e, tau, beta);
      t[3] = beta;
      reflect(t + 4, 3, 1, 2, 2, e, tau, true);   // rows 1..2 of columns 1..2
      reflect(t + 1, 1, 3, 2, 3, e, tau, false);  // columns 1..2 of rows 0..2
      reflect(u + 4, 3, 1, 2, 2, e, tau, true);
      t[6] = 0.0f;
      // (the second step of the reduction reflects a single element: tau = 0, nothing changes)
    }
    // real Schur form by shifted QR steps on the active window [lo, hi]
    int hi = 2, it = 0, total = 0;
    float exshift = 0.0f, nrm = 0.0f;
    for (int j = 0; j < 3; j++)
      for (int i = 0; i < (j + 2 < 3 ? j + 2 : 3); i++)
        nrm += fabs_(t[3 * i + j]);
    while (nrm != 0.0f && hi >= 0)
    {
      int lo = hi;
      for (; lo > 0; lo--)
      {
        const float sdiag = fmax_((fabs_(t[4 * (lo - 1)]) + fabs_(t[4 * lo])) * eps, FLT_MIN);
        if (fabs_(t[3 * lo + lo - 1]) <= sdiag)
          break;
      }
      if (lo == hi)
      {
        t[4 * hi] = t[4 * hi] + exshift;
        if (hi > 0)
          t[3 * hi + hi - 1] = 0.0f;
        hi--;
        it = 0;
        continue;
      }
      if (lo == hi - 1)
      {
        float* d0 = &t[4 * (hi - 1)];
        float* d1 = &t[4 * hi];
        const float p = 0.5f * (*d0 - *d1);
        const float q = p * p + t[3 * hi + hi - 1] * t[3 * (hi - 1) + hi];
        *d1 += exshift;
        *d0 += exshift;
        if (q >= 0.0f)
        {
          const float z = sqrtf(fabs_(q));
          float c, s;
          givens(p >= 0.0f ? p + z : p - z, t[3 * hi + hi - 1], c, s);
          rotate(&t[3 * (hi - 1) + hi - 1], &t[3 * hi + hi - 1], 1, 3 - (hi - 1), c, s);  // rows hi-1, hi from column hi-1 on
          rotate(&t[hi - 1], &t[hi], 3, hi + 1, c, s);                                    // columns hi-1, hi of rows 0..hi
          t[3 * hi + hi - 1] = 0.0f;
          rotate(&u[hi - 1], &u[hi], 3, 3, c, s);
        }
        if (hi > 1)
          t[3 * (hi - 1) + hi - 2] = 0.0f;
        hi -= 2;
        it = 0;
        continue;
      }
      // window = the whole matrix (lo = 0, hi = 2)
      float sh0 = t[4 * hi], sh1 = t[4 * (hi - 1)], sh2 = t[3 * hi + hi - 1] * t[3 * (hi - 1) + hi];
      if (it == 10)
      {
        exshift += sh0;
        for (int i = 0; i <= hi; i++)
          t[4 * i] -= sh0;
        const float s = fabs_(t[3 * hi + hi - 1]) + fabs_(t[3 * (hi - 1) + hi - 2]);
        sh0 = 0.75f * s;
        sh1 = 0.75f * s;
        sh2 = -0.4375f * s * s;
      }
      if (it == 30)
      {
        float s = (sh1 - sh0) / 2.0f;
        s = s * s + sh2;
        if (s > 0.0f)
        {
          s = sqrtf(s);
          if (sh1 < sh0)
            s = -s;
          s = s + (sh1 - sh0) / 2.0f;
          s = sh0 - sh2 / s;
          exshift += s;
          for (int i = 0; i <= hi; i++)
            t[4 * i] -= s;
          sh0 = sh1 = sh2 = 0.964f;
        }
      }
      it++;
      if (++total > 120)
        break;
      // first column of the double-shift polynomial at row m (the search for a smaller start row ends at lo for n = 3)
      const int m = hi - 2;
      float x[3];
      {
        const float tmm = t[4 * m], r = sh0 - tmm, s = sh1 - tmm;
        x[0] = (r * s - sh2) / t[3 * (m + 1) + m] + t[3 * m + m + 1];
        x[1] = t[4 * (m + 1)] - tmm - r - s;
        x[2] = t[3 * (m + 2) + m + 1];
      }
      {
        float e[2], tau, beta;
        reflector(x, 3, e, tau, beta);
        if (beta != 0.0f)
        {
          reflect(t + 4 * m, 3, 1, 3, 3 - m, e, tau, true);                     // rows m..m+2, columns m..2
          reflect(t + m, 1, 3, 3, (hi < m + 3 ? hi : m + 3) + 1, e, tau, false);  // columns m..m+2, rows 0..min(hi, m+3)
          reflect(u + m, 1, 3, 3, 3, e, tau, false);
        }
      }
      {
        const float y[2] = {t[3 * (hi - 1) + hi - 2], t[3 * hi + hi - 2]};
        float e[1], tau, beta;
        reflector(y, 2, e, tau, beta);
        if (beta != 0.0f)
        {
          t[3 * (hi - 1) + hi - 2] = beta;
          reflect(t + 4 * (hi - 1), 3, 1, 2, 3 - hi + 1, e, tau, true);  // rows hi-1..hi, columns hi-1..2
          reflect(t + hi - 1, 1, 3, 2, hi + 1, e, tau, false);           // columns hi-1..hi, rows 0..hi
          reflect(u + hi - 1, 1, 3, 2, 3, e, tau, false);
        }
      }
      t[3 * hi + hi - 2] = 0.0f;  // round-off below the sub-diagonal
    }
    for (int i = 0; i < 9; i++)
      t[i] *= big;
  }
  // eigenvalues of the quasi-triangular form
  float im[3];
  for (int i = 0; i < 3;)
  {
    if (i == 2 || t[3 * (i + 1) + i] == 0.0f)
    {
      val[i] = t[4 * i];
      im[i] = 0.0f;
      i++;
      continue;
    }
    const float p = 0.5f * (t[4 * i] - t[4 * (i + 1)]);
    float b = t[3 * (i + 1) + i], c = t[3 * i + i + 1];
    const float mx = fmax_(fabs_(p), fmax_(fabs_(b), fabs_(c)));
    b /= mx;
    c /= mx;
    const float p0 = p / mx;
    const float z = mx * sqrtf(fabs_(p0 * p0 + b * c));
    val[i] = val[i + 1] = t[4 * (i + 1)] + p;
    im[i] = z;
    im[i + 1] = -z;
    i += 2;
  }
  // eigenvectors of the triangular form by back substitution, then of the input through the Schur vectors
  float nrm = 0.0f;
  for (int j = 0; j < 3; j++)
    for (int c = (j > 0 ? j - 1 : 0); c < 3; c++)
      nrm += fabs_(t[3 * j + c]);
  if (nrm != 0.0f)
  {
    for (int n = 2; n >= 0; n--)
    {
      const float p = val[n], q = im[n];
      if (q == 0.0f)
      {
        float lastr = 0.0f, lastw = 0.0f;
        int l = n;
        t[4 * n] = 1.0f;
        for (int k = n - 1; k >= 0; k--)
        {
          const float w = t[4 * k] - p;
          float r = 0.0f;
          for (int c = l; c <= n; c++)
            r += t[3 * k + c] * t[3 * c + n];
          if (im[k] < 0.0f)
          {
            lastw = w;
            lastr = r;
            continue;
          }
          l = k;
          if (im[k] == 0.0f)
            t[3 * k + n] = w != 0.0f ? -r / w : -r / (eps * nrm);
          else
          {
            const float xx = t[3 * k + k + 1], yy = t[3 * (k + 1) + k];
            const float den = (val[k] - p) * (val[k] - p) + im[k] * im[k];
            const float tt = (xx * lastr - lastw * r) / den;
            t[3 * k + n] = tt;
            t[3 * (k + 1) + n] = fabs_(xx) > fabs_(lastw) ? (-r - w * tt) / xx : (-lastr - yy * tt) / lastw;
          }
          const float a = fabs_(t[3 * k + n]);
          if ((eps * a) * a > 1.0f)
            for (int r2 = k; r2 < 3; r2++)
              t[3 * r2 + n] /= a;
        }
      }
      else if (q < 0.0f && n > 0)
      {
        // complex pair in columns n-1, n (rounding noise on a repeated eigenvalue can produce one)
        int l = n - 1;
        if (fabs_(t[3 * n + n - 1]) > fabs_(t[3 * (n - 1) + n]))
        {
          t[4 * (n - 1)] = q / t[3 * n + n - 1];
          t[3 * (n - 1) + n] = -(t[4 * n] - p) / t[3 * n + n - 1];
        }
        else
        {
          const float bi = -t[3 * (n - 1) + n], cr = t[4 * (n - 1)] - p, den = cr * cr + q * q;
          t[4 * (n - 1)] = (0.0f * cr + bi * q) / den;
          t[3 * (n - 1) + n] = (bi * cr - 0.0f * q) / den;
        }
        t[3 * n + n - 1] = 0.0f;
        t[4 * n] = 1.0f;
        for (int k = n - 2; k >= 0; k--)
        {
          float ra = 0.0f, sa = 0.0f;
          for (int c = l; c <= n; c++)
          {
            ra += t[3 * k + c] * t[3 * c + n - 1];
            sa += t[3 * k + c] * t[3 * c + n];
          }
          const float w = t[4 * k] - p;
          if (im[k] < 0.0f)
            continue;
          l = k;
          if (im[k] == 0.0f)
          {
            const float ar = -ra, ai = -sa, den = w * w + q * q;
            t[3 * k + n - 1] = (ar * w + ai * q) / den;
            t[3 * k + n] = (ai * w - ar * q) / den;
          }
          const float a = fmax_(fabs_(t[3 * k + n - 1]), fabs_(t[3 * k + n]));
          if ((eps * a) * a > 1.0f)
            for (int r2 = k; r2 < 3; r2++)
            {
              t[3 * r2 + n - 1] /= a;
              t[3 * r2 + n] /= a;
            }
        }
        n--;
      }
    }
    for (int j = 2; j >= 0; j--)
    {
      float col[3];
      for (int r = 0; r < 3; r++)
      {
        float acc = 0.0f;
        for (int c = 0; c <= j; c++)
          acc += u[3 * r + c] * t[3 * c + j];
        col[r] = acc;
      }
      for (int r = 0; r < 3; r++)
        u[3 * r + j] = col[r];
    }
  }
  for (int j = 0; j < 3; j++)
  {
    if (fabs_(im[j]) <= fabs_(val[j]) * (2.0f * eps) || j == 2)
    {
      const float nn = sqrtf(u[j] * u[j] + u[3 + j] * u[3 + j] + u[6 + j] * u[6 + j]);
      for (int r = 0; r < 3; r++)
        vec[r][j] = u[3 * r + j] / nn;
      continue;
    }
    float sq = 0.0f;
    for (int r = 0; r < 3; r++)
      sq += u[3 * r + j] * u[3 * r + j] + u[3 * r + j + 1] * u[3 * r + j + 1];
    const float nn = sqrtf(sq);
    for (int r = 0; r < 3; r++)
      vec[r][j] = vec[r][j + 1] = u[3 * r + j] / nn;
    j++;
  }
}

}  // namespace ve
