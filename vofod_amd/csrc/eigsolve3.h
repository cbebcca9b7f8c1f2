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
// The matrices are 3 x 3 and every index below is a compile-time constant after unrolling (the window of the QR iteration is
// a three-way case, reflections and rotations are templates over their rows / columns): on the device the 18 matrix elements
// live in registers.  A first version with run-time strides kept them in scratch memory and made the tail's gate kernel 3 x
// slower - enough to turn the tail stream into the pipeline's bottleneck.
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

// Householder vector of (x0, x1[, x2]): -> (beta, 0, ..), essential part e, factor tau (Eigen makeHouseholder)
template <int N>
VE_HD inline void reflector(const float (&x)[N], float (&e)[N - 1], float& tau, float& beta)
{
  float tail = 0.0f;
#pragma unroll
  for (int i = 1; i < N; i++)
    tail += x[i] * x[i];
  const float head = x[0];
  if (tail <= FLT_MIN)
  {
    tau = 0.0f;
    beta = head;
#pragma unroll
    for (int i = 0; i < N - 1; i++)
      e[i] = 0.0f;
    return;
  }
  beta = sqrtf(head * head + tail);
  if (head >= 0.0f)
    beta = -beta;
#pragma unroll
  for (int i = 0; i < N - 1; i++)
    e[i] = x[i + 1] / (head - beta);
  tau = (beta - head) / beta;
}

// H = I - tau (1, e)(1, e)^T applied from the left to rows R0 .. R0 + N - 1 of the columns C0 .. C0 + NC - 1 of m (row major 3 x 3):
// Eigen's applyHouseholderOnTheLeft, association (tau * e_i) * tmp
template <int R0, int N, int C0, int NC>
VE_HD inline void reflect_rows(float (&m)[9], const float (&e)[N - 1], float tau)
{
  if (tau == 0.0f)
    return;
#pragma unroll
  for (int c = C0; c < C0 + NC; c++)
  {
    float t = 0.0f;
#pragma unroll
    for (int i = 1; i < N; i++)
      t += e[i - 1] * m[3 * (R0 + i) + c];
    t += m[3 * R0 + c];
    m[3 * R0 + c] -= tau * t;
#pragma unroll
    for (int i = 1; i < N; i++)
      m[3 * (R0 + i) + c] -= (tau * e[i - 1]) * t;
  }
}

// ... from the right to the columns C0 .. C0 + N - 1 of the rows 0 .. NR - 1: applyHouseholderOnTheRight, (tau * tmp) * e_i
template <int C0, int N, int NR>
VE_HD inline void reflect_cols(float (&m)[9], const float (&e)[N - 1], float tau)
{
  if (tau == 0.0f)
    return;
#pragma unroll
  for (int r = 0; r < NR; r++)
  {
    float t = 0.0f;
#pragma unroll
    for (int i = 1; i < N; i++)
      t += m[3 * r + C0 + i] * e[i - 1];
    t += m[3 * r + C0];
    m[3 * r + C0] -= tau * t;
#pragma unroll
    for (int i = 1; i < N; i++)
      m[3 * r + C0 + i] -= (tau * t) * e[i - 1];
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

VE_HD inline void rot(float& x, float& y, float c, float s)
{
  const float a = x, b = y;
  x = c * a - s * b;
  y = s * a + c * b;
}

// RealSchur::splitOffTwoRows for the rows / columns (HI - 1, HI)
template <int HI>
VE_HD inline void split_two(float (&t)[9], float (&u)[9], float exshift)
{
  constexpr int A = HI - 1, B = HI;
  const float p = 0.5f * (t[4 * A] - t[4 * B]);
  const float q = p * p + t[3 * B + A] * t[3 * A + B];
  t[4 * B] += exshift;
  t[4 * A] += exshift;
  if (q >= 0.0f)
  {
    const float z = sqrtf(fabs_(q));
    float c, s;
    givens(p >= 0.0f ? p + z : p - z, t[3 * B + A], c, s);
#pragma unroll
    for (int col = A; col < 3; col++)  // rows A, B from column A on
      rot(t[3 * A + col], t[3 * B + col], c, s);
#pragma unroll
    for (int row = 0; row <= B; row++)  // columns A, B of the rows 0 .. B
      rot(t[3 * row + A], t[3 * row + B], c, s);
    t[3 * B + A] = 0.0f;
#pragma unroll
    for (int row = 0; row < 3; row++)
      rot(u[3 * row + A], u[3 * row + B], c, s);
  }
  if constexpr (HI > 1)
    t[3 * A + A - 1] = 0.0f;
}

// Eigenvalues (real parts, in the solver's order) and the real parts of the normalised eigenvectors (column j of vec) of a
// real 3 x 3 matrix, as Eigen::EigenSolver<Matrix3f>(cov).eigenvalues().real() / .eigenvectors().real() deliver them.
VE_HD inline void eigsolve3(const float cov[3][3], float val[3], float vec[3][3])
{
  const float eps = FLT_EPSILON;
  float t[9], u[9] = {1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f};
  float big = 0.0f;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      big = fmax_(big, fabs_(cov[i][j]));
#pragma unroll
  for (int i = 0; i < 9; i++)
    t[i] = 0.0f;
  if (!(big < FLT_MIN))
  {
    // Hessenberg form of cov / big: one reflector on (a10, a20), applied from both sides; Q is that reflector
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++)
        t[3 * i + j] = cov[i][j] / big;
    {
      const float x[2] = {t[3], t[6]};
      float e[1], tau, beta;
      reflector<2>(x, e, tau, beta);
      t[3] = beta;
      reflect_rows<1, 2, 1, 2>(t, e, tau);  // rows 1..2 of columns 1..2
      reflect_cols<1, 2, 3>(t, e, tau);     // columns 1..2 of rows 0..2
      reflect_rows<1, 2, 1, 2>(u, e, tau);
      t[6] = 0.0f;
      // (the second step of the reduction reflects a single element: tau = 0, nothing changes)
    }
    // real Schur form by shifted QR steps on the active window [lo, hi]
    int hi = 2, it = 0, total = 0;
    float exshift = 0.0f, nrm = 0.0f;
    nrm = fabs_(t[0]) + fabs_(t[3]);                          // column 0: rows 0..1
    nrm += fabs_(t[1]) + fabs_(t[4]) + fabs_(t[7]);           // column 1
    nrm += fabs_(t[2]) + fabs_(t[5]) + fabs_(t[8]);           // column 2
    auto small_sub = [&](float d0, float d1, float sub) { return fabs_(sub) <= fmax_((fabs_(d0) + fabs_(d1)) * eps, FLT_MIN); };
    while (nrm != 0.0f && hi >= 0)
    {
      if (hi == 2)
      {
        if (small_sub(t[4], t[8], t[7]))  // lo == hi: one root
        {
          t[8] = t[8] + exshift;
          t[7] = 0.0f;
          hi = 1;
          it = 0;
          continue;
        }
        if (small_sub(t[0], t[4], t[3]))  // lo == hi - 1: two roots
        {
          split_two<2>(t, u, exshift);
          hi = 0;
          it = 0;
          continue;
        }
        // window = the whole matrix
        float sh0 = t[8], sh1 = t[4], sh2 = t[7] * t[5];
        if (it == 10)
        {
          exshift += sh0;
          t[0] -= sh0;
          t[4] -= sh0;
          t[8] -= sh0;
          const float s = fabs_(t[7]) + fabs_(t[3]);
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
            t[0] -= s;
            t[4] -= s;
            t[8] -= s;
            sh0 = sh1 = sh2 = 0.964f;
          }
        }
        it++;
        if (++total > 120)
          break;
        // first column of the double-shift polynomial at row 0 (the search for a later start row ends at lo for n = 3)
        float x[3];
        {
          const float tmm = t[0], r = sh0 - tmm, s = sh1 - tmm;
          x[0] = (r * s - sh2) / t[3] + t[1];
          x[1] = t[4] - tmm - r - s;
          x[2] = t[7];
        }
        {
          float e[2], tau, beta;
          reflector<3>(x, e, tau, beta);
          if (beta != 0.0f)
          {
            reflect_rows<0, 3, 0, 3>(t, e, tau);
            reflect_cols<0, 3, 3>(t, e, tau);
            reflect_cols<0, 3, 3>(u, e, tau);
          }
        }
        {
          const float y[2] = {t[3], t[6]};
          float e[1], tau, beta;
          reflector<2>(y, e, tau, beta);
          if (beta != 0.0f)
          {
            t[3] = beta;
            reflect_rows<1, 2, 1, 2>(t, e, tau);  // rows 1..2, columns 1..2
            reflect_cols<1, 2, 3>(t, e, tau);     // columns 1..2, rows 0..2
            reflect_cols<1, 2, 3>(u, e, tau);
          }
        }
        t[6] = 0.0f;  // round-off below the sub-diagonal
      }
      else if (hi == 1)
      {
        if (small_sub(t[0], t[4], t[3]))
        {
          t[4] = t[4] + exshift;
          t[3] = 0.0f;
          hi = 0;
          it = 0;
          continue;
        }
        split_two<1>(t, u, exshift);
        hi = -1;
        it = 0;
      }
      else
      {
        t[0] = t[0] + exshift;
        hi = -1;
        it = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < 9; i++)
      t[i] *= big;
  }
  // eigenvalues of the quasi-triangular form: 2 x 2 blocks can sit at (0,1) or (1,2)
  float im[3] = {0.0f, 0.0f, 0.0f};
  auto pair_values = [&](float d0, float d1, float sub, float sup, float& re, float& z) {
    const float p = 0.5f * (d0 - d1);
    float b = sub, c = sup;
    const float mx = fmax_(fabs_(p), fmax_(fabs_(b), fabs_(c)));
    b /= mx;
    c /= mx;
    const float p0 = p / mx;
    z = mx * sqrtf(fabs_(p0 * p0 + b * c));
    re = d1 + p;
  };
  if (t[3] != 0.0f)
  {
    float re, z;
    pair_values(t[0], t[4], t[3], t[1], re, z);
    val[0] = val[1] = re;
    im[0] = z;
    im[1] = -z;
    val[2] = t[8];
  }
  else
  {
    val[0] = t[0];
    if (t[7] != 0.0f)
    {
      float re, z;
      pair_values(t[4], t[8], t[7], t[5], re, z);
      val[1] = val[2] = re;
      im[1] = z;
      im[2] = -z;
    }
    else
    {
      val[1] = t[4];
      val[2] = t[8];
    }
  }
  // eigenvectors of the triangular form by back substitution, then of the input through the Schur vectors
  float nrm2 = 0.0f;  // (summed element by element, row 0 from column 0, row 1 from column 0, row 2 from column 1: it enters a quotient below)
  nrm2 += fabs_(t[0]);
  nrm2 += fabs_(t[1]);
  nrm2 += fabs_(t[2]);
  nrm2 += fabs_(t[3]);
  nrm2 += fabs_(t[4]);
  nrm2 += fabs_(t[5]);
  nrm2 += fabs_(t[7]);
  nrm2 += fabs_(t[8]);
  if (nrm2 != 0.0f)
  {
    bool skip_next = false;  // a complex pair takes two columns
#pragma unroll
    for (int n = 2; n >= 0; n--)
    {
      if (skip_next)
      {
        skip_next = false;
        continue;
      }
      const float p = val[n], q = im[n];
      if (q == 0.0f)
      {
        float lastr = 0.0f, lastw = 0.0f;
        int l = n;
        t[4 * n] = 1.0f;
#pragma unroll
        for (int k = 2; k >= 0; k--)
        {
          if (k >= n)
            continue;
          const float w = t[4 * k] - p;
          float r = 0.0f;
#pragma unroll
          for (int c = 0; c < 3; c++)
            if (c >= l && c <= n)
              r += t[3 * k + c] * t[3 * c + n];
          if (im[k] < 0.0f)
          {
            lastw = w;
            lastr = r;
            continue;
          }
          l = k;
          if (im[k] == 0.0f)
            t[3 * k + n] = w != 0.0f ? -r / w : -r / (eps * nrm2);
          else
          {
            if constexpr (true)
            {
              // (k + 1 <= 2 whenever a pair starts at k)
              const int k1 = k < 2 ? k + 1 : 2;
              float xx = 0.0f, yy = 0.0f;
#pragma unroll
              for (int c = 0; c < 3; c++)
                if (c == k1)
                {
                  xx = t[3 * k + c];
                  yy = t[3 * c + k];
                }
              const float den = (val[k] - p) * (val[k] - p) + im[k] * im[k];
              const float tt = (xx * lastr - lastw * r) / den;
              t[3 * k + n] = tt;
              const float other = fabs_(xx) > fabs_(lastw) ? (-r - w * tt) / xx : (-lastr - yy * tt) / lastw;
#pragma unroll
              for (int c = 0; c < 3; c++)
                if (c == k1)
                  t[3 * c + n] = other;
            }
          }
          const float a = fabs_(t[3 * k + n]);
          if ((eps * a) * a > 1.0f)
          {
#pragma unroll
            for (int r2 = 0; r2 < 3; r2++)
              if (r2 >= k)
                t[3 * r2 + n] /= a;
          }
        }
      }
      else if (q < 0.0f && n > 0)
      {
        // complex pair in columns n-1, n (rounding noise on a repeated eigenvalue can produce one)
        constexpr int dummy = 0;
        (void)dummy;
        if (n == 2)
        {
          // columns 1, 2
          if (fabs_(t[7]) > fabs_(t[5]))
          {
            t[4] = q / t[7];
            t[5] = -(t[8] - p) / t[7];
          }
          else
          {
            const float bi = -t[5], cr = t[4] - p, den = cr * cr + q * q;
            t[4] = (0.0f * cr + bi * q) / den;
            t[5] = (bi * cr - 0.0f * q) / den;
          }
          t[7] = 0.0f;
          t[8] = 1.0f;
          {
            // k = 0, l = 1
            float ra = 0.0f, sa = 0.0f;
            ra += t[1] * t[4];
            sa += t[1] * t[5];
            ra += t[2] * t[7];
            sa += t[2] * t[8];
            const float w = t[0] - p;
            if (!(im[0] < 0.0f))
            {
              if (im[0] == 0.0f)
              {
                const float ar = -ra, ai = -sa, den = w * w + q * q;
                t[1] = (ar * w + ai * q) / den;
                t[2] = (ai * w - ar * q) / den;
              }
              const float a = fmax_(fabs_(t[1]), fabs_(t[2]));
              if ((eps * a) * a > 1.0f)
              {
                t[1] /= a;
                t[2] /= a;
                t[4] /= a;
                t[5] /= a;
                t[7] /= a;
                t[8] /= a;
              }
            }
          }
        }
        else
        {
          // columns 0, 1 (n == 1): nothing above them
          if (fabs_(t[3]) > fabs_(t[1]))
          {
            t[0] = q / t[3];
            t[1] = -(t[4] - p) / t[3];
          }
          else
          {
            const float bi = -t[1], cr = t[0] - p, den = cr * cr + q * q;
            t[0] = (0.0f * cr + bi * q) / den;
            t[1] = (bi * cr - 0.0f * q) / den;
          }
          t[3] = 0.0f;
          t[4] = 1.0f;
        }
        skip_next = true;
      }
    }
#pragma unroll
    for (int j = 2; j >= 0; j--)
    {
      float col[3];
#pragma unroll
      for (int r = 0; r < 3; r++)
      {
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; c++)
          if (c <= j)
            acc += u[3 * r + c] * t[3 * c + j];
        col[r] = acc;
      }
#pragma unroll
      for (int r = 0; r < 3; r++)
        u[3 * r + j] = col[r];
    }
  }
  bool pair_done = false;
#pragma unroll
  for (int j = 0; j < 3; j++)
  {
    if (pair_done)
    {
      pair_done = false;
      continue;
    }
    if (fabs_(im[j]) <= fabs_(val[j]) * (2.0f * eps) || j == 2)
    {
      const float nn = sqrtf(u[j] * u[j] + u[3 + j] * u[3 + j] + u[6 + j] * u[6 + j]);
#pragma unroll
      for (int r = 0; r < 3; r++)
        vec[r][j] = u[3 * r + j] / nn;
      continue;
    }
    if constexpr (true)
    {
      // (j <= 1 here)
      float sq = 0.0f;
#pragma unroll
      for (int r = 0; r < 3; r++)
      {
        const float a = u[3 * r + j], b = j < 2 ? u[3 * r + (j < 2 ? j + 1 : 2)] : 0.0f;
        sq += a * a + b * b;
      }
      const float nn = sqrtf(sq);
#pragma unroll
      for (int r = 0; r < 3; r++)
      {
        const float v = u[3 * r + j] / nn;
        vec[r][j] = v;
        if (j < 2)
          vec[r][j < 2 ? j + 1 : 2] = v;
      }
      pair_done = true;
    }
  }
}

}  // namespace ve
