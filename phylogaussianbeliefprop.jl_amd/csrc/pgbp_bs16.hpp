// BS16: symmetric block-packed device layout of beliefs of dimension 16 or 32 (and of the residuals of
// 16-dimensional sepsets), used while every scheduled task runs on the register-resident kernel.
//
// A precision matrix is symmetric, and the fast kernel's lane (a, b) owns the 2 x 2 block
// {2a, 2a+1} x {2b, 2b+1} of every 16 x 16 tile.  BS16 stores, per tile,
//   * symmetric (diagonal) tile: the 36 blocks with a <= b, block k = b(b+1)/2 + a, 4 doubles each
//     [T(2a,2b), T(2a+1,2b), T(2a,2b+1), T(2a+1,2b+1)]                         -> 144 doubles
//   * off-diagonal tile T10 (rows 16..31, cols 0..15): all 64 blocks, block a + 8b -> 256 doubles
// so that one wave-instruction of 32 B per lane reads or writes a whole tile's stored part contiguously.
//   dim 16 record: [T00 sym 144 | h 16 | g]                        = 161 doubles (plain: 273)
//   dim 32 record: [T00 sym 144 | T10 256 | T11 sym 144 | h 32 | g] = 577 doubles (plain: 1057)
//   residual (s = 16): [dJ sym 144 | dh 16]                         = 160 doubles (plain: 272)
// Records keep their plain-layout offsets (and size); only the first part is used.  Beliefs of any other
// dimension keep the plain layout.  The lower triangle is implied by symmetry: plain -> BS16 keeps the upper
// triangle (what PDMat(Symmetric(J)) reads, src/beliefupdates.jl:68), BS16 -> plain mirrors it.
#pragma once
#include <cstdint>

namespace pgbp {
namespace bs16 {

constexpr int kSym = 144;   // doubles of a packed symmetric 16 x 16 tile
constexpr int kFull = 256;  // doubles of a full 16 x 16 tile
constexpr int kT00 = 0, kT10 = 144, kT11 = 400, kH32 = 544, kG32 = 576, kLen32 = 577;
constexpr int kH16 = 144, kG16 = 160, kLen16 = 161;
constexpr int kResH = 144, kResLen = 160;

__host__ __device__ inline bool applies(int m) { return m == 16 || m == 32; }

// offset, inside a packed symmetric tile, of element (r, c), 0 <= r, c < 16 (either triangle)
__host__ __device__ inline int sym_off(int r, int c) {
  int a = r >> 1, b = c >> 1;
  if (a > b) {  // stored through the transposed block
    const int t = r; r = c; c = t;
    a = r >> 1; b = c >> 1;
  }
  return (b * (b + 1) / 2 + a) * 4 + (r & 1) + 2 * (c & 1);
}
// offset inside the full tile T10 of element (r, c) (local indices)
__host__ __device__ inline int full_off(int r, int c) { return ((r >> 1) + 8 * (c >> 1)) * 4 + (r & 1) + 2 * (c & 1); }

// offset in a BS16 record of dimension m (16 or 32) of J(r, c)
__host__ __device__ inline int J_off(int m, int r, int c) {
  if (m == 16) return sym_off(r, c);
  const int R = r >> 4, C = c >> 4;
  if (R == C) return (R == 0 ? kT00 : kT11) + sym_off(r & 15, c & 15);
  return R > C ? kT10 + full_off(r & 15, c & 15) : kT10 + full_off(c & 15, r & 15);
}
__host__ __device__ inline int h_off(int m, int r) { return (m == 16 ? kH16 : kH32) + r; }
__host__ __device__ inline int g_off(int m) { return m == 16 ? kG16 : kG32; }
// is (r, c) the stored representative of its symmetric pair?  (every stored slot is visited exactly once)
__host__ __device__ inline bool canonical(int m, int r, int c) {
  const int R = r >> 4, C = c >> 4;
  if (R != C) return R > C;
  return ((r & 15) >> 1) <= ((c & 15) >> 1);
}

}  // namespace bs16
}  // namespace pgbp
