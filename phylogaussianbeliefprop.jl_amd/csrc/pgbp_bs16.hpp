// BSP ("BS16" for P = 16): symmetric block-packed device layout of beliefs of dimension P or 2P (and of the
// residuals of P-dimensional sepsets), used while every scheduled task runs on the register-resident kernel.
// P = fast-class sepset dimension (even, <= 16); G = P / 2 = side of the kernel's lane grid.
//
// A precision matrix is symmetric, and the fast kernel's lane (a, b), a, b < G, owns the 2 x 2 block
// {2a, 2a+1} x {2b, 2b+1} of every P x P tile.  The layout stores, per tile,
//   * symmetric (diagonal) tile: the G(G+1)/2 blocks with a <= b, block k = b(b+1)/2 + a, 4 doubles each
//     [T(2a,2b), T(2a+1,2b), T(2a,2b+1), T(2a+1,2b+1)]                 -> sym(P) = 2G(G+1) doubles (144 for P = 16)
//   * off-diagonal tile T10 (rows P..2P-1, cols 0..P-1): all G*G blocks, block a + G*b -> P*P doubles
// so that one wave-instruction of 32 B per lane reads or writes a whole tile's stored part contiguously.
//   dim P record : [T00 sym | h P | g]                        (P = 16: 161 doubles, plain 273)
//   dim 2P record: [T00 sym | T10 P*P | T11 sym | h 2P | g]   (P = 16: 577 doubles, plain 1057)
//   residual (s = P): [dJ sym | dh P]                          (P = 16: 160 doubles, plain 272)
// Records keep their plain-layout offsets (and size); only the first part is used.  Beliefs of any other
// dimension keep the plain layout.  The lower triangle is implied by symmetry: plain -> packed keeps the upper
// triangle (what PDMat(Symmetric(J)) reads, src/beliefupdates.jl:68), packed -> plain mirrors it.
#pragma once
#include <cstdint>

namespace pgbp {
namespace bs16 {

__host__ __device__ constexpr int sym_len(int P) { return (P / 2) * (P / 2 + 1) * 2; }
__host__ __device__ constexpr int full_len(int P) { return P * P; }
__host__ __device__ constexpr int t10(int P) { return sym_len(P); }
__host__ __device__ constexpr int t11(int P) { return sym_len(P) + full_len(P); }
__host__ __device__ constexpr int h2(int P) { return 2 * sym_len(P) + full_len(P); }  // h of a 2P record
__host__ __device__ constexpr int g2(int P) { return h2(P) + 2 * P; }
__host__ __device__ constexpr int len2(int P) { return g2(P) + 1; }
__host__ __device__ constexpr int h1(int P) { return sym_len(P); }                    // h of a P record
__host__ __device__ constexpr int g1(int P) { return sym_len(P) + P; }
__host__ __device__ constexpr int len1(int P) { return g1(P) + 1; }

__host__ __device__ inline bool applies(int m, int P) { return P > 0 && (m == P || m == 2 * P); }

// offset, inside a packed symmetric tile, of element (r, c), 0 <= r, c < P (either triangle)
__host__ __device__ inline int sym_off(int r, int c) {
  int a = r >> 1, b = c >> 1;
  if (a > b) {  // stored through the transposed block
    const int t = r; r = c; c = t;
    a = r >> 1; b = c >> 1;
  }
  return (b * (b + 1) / 2 + a) * 4 + (r & 1) + 2 * (c & 1);
}
// offset inside the full tile T10 of element (r, c) (local indices)
__host__ __device__ inline int full_off(int r, int c, int P) {
  return ((r >> 1) + (P / 2) * (c >> 1)) * 4 + (r & 1) + 2 * (c & 1);
}

// offset in a packed record of dimension m (P or 2P) of J(r, c)
__host__ __device__ inline int J_off(int m, int r, int c, int P) {
  if (m == P) return sym_off(r, c);
  const int R = r >= P, C = c >= P;
  const int lr = r - R * P, lc = c - C * P;
  if (R == C) return (R == 0 ? 0 : t11(P)) + sym_off(lr, lc);
  return R > C ? t10(P) + full_off(lr, lc, P) : t10(P) + full_off(lc, lr, P);
}
__host__ __device__ inline int h_off(int m, int r, int P) { return (m == P ? h1(P) : h2(P)) + r; }
__host__ __device__ inline int g_off(int m, int P) { return m == P ? g1(P) : g2(P); }
__host__ __device__ inline int rec_len(int m, int P) { return m == P ? len1(P) : len2(P); }
// is (r, c) the stored representative of its symmetric pair?  (every stored slot is visited exactly once)
__host__ __device__ inline bool canonical(int m, int r, int c, int P) {
  const int R = (m != P) && r >= P, C = (m != P) && c >= P;
  if (R != C) return R > C;
  return ((r - R * P) >> 1) <= ((c - C * P) >> 1);
}

}  // namespace bs16
}  // namespace pgbp
