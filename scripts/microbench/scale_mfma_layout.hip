// One-hot probe of v_mfma_scale_f32_32x32x64_f8f6f4 (gfx950): which (row, k) each operand byte (lane, byte 0..31) is,
// and which lane's E8M0 scale byte multiplies it.  A position pair (A byte, B byte) contributes to D iff their k agree.
// hipcc --offload-arch=gfx950 -O3 -o layout scale_mfma_layout.hip && ./layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float float16v __attribute__((ext_vector_type(16)));
typedef int int8v __attribute__((ext_vector_type(8)));

// block = one A position (la, ba); loops over all B positions; out[apos][bpos] = (lane << 8 | reg) + 1 of the nonzero D entry, 0 = none
__global__ void onehot(unsigned short* out) {
  const int lane = threadIdx.x, apos = blockIdx.x, la = apos >> 5, ba = apos & 31;
  int8v a = {};
  if (lane == la) a[ba >> 2] = 0x38 << (8 * (ba & 3));       // e4m3 1.0
  for (int bpos = 0; bpos < 2048; ++bpos) {
    const int lb = bpos >> 5, bb = bpos & 31;
    int8v b = {};
    if (lane == lb) b[bb >> 2] = 0x3c << (8 * (bb & 3));     // e5m2 1.0
    float16v acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    int hit = 0;
    for (int r = 0; r < 16; ++r) if (acc[r] != 0.f) hit = ((lane << 8) | r) + 1;
    // at most one lane has a hit
    unsigned long long m = __ballot(hit != 0);
    if (m) { if (hit) out[(size_t)apos * 2048 + bpos] = (unsigned short)hit; }
    else if (lane == 0) out[(size_t)apos * 2048 + bpos] = 0;
  }
}

// scale ownership: A = all ones, B = all ones; scale of A set to 2.0 (128) in ONE lane ls only; D[i][j] = 64 + (#k of row i under that scale)
__global__ void scale_owner(float* out, int which, int dhalf) {
  const int lane = threadIdx.x, ls = blockIdx.x;
  int8v a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x3c3c3c3c; }
  // additionally zero all but ONE byte position of A per launch to see which bytes a lane's scale covers: done on the host by sweeping `which`
  int8v a1 = {};
  if ((lane >> 5) == dhalf) a1[which >> 2] = 0x38 << (8 * (which & 3));     // only byte `which` of the lanes of one half-wave is 1.0
  const int sa = lane == ls ? 0x80 : 0x7f;
  float16v acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(which >= 0 ? a1 : a, b, acc, 0, 1, 0, sa, 0, 0x7f7f7f7f);
  for (int r = 0; r < 16; ++r) out[((size_t)ls * 64 + lane) * 16 + r] = acc[r];
}

int main() {
  unsigned short* d; hipMalloc(&d, 2048ull * 2048 * 2); hipMemset(d, 0xff, 2048ull * 2048 * 2);
  onehot<<<2048, 64>>>(d);
  std::vector<unsigned short> h(2048ull * 2048);
  hipMemcpy(h.data(), d, h.size() * 2, hipMemcpyDeviceToHost);
  // k classes: A positions that match the same set of B (lane&..)... derive: for A pos, list of matching B pos
  // 1) row of an A position = D row of its hits; 2) k id: assign by first matching B byte index pattern
  std::vector<int> a_row(2048, -1), a_kid(2048, -1), b_col(2048, -1), b_kid(2048, -1);
  int nk = 0;
  for (int ap = 0; ap < 2048; ++ap) {
    int cnt = 0;
    for (int bp = 0; bp < 2048; ++bp) {
      int v = h[(size_t)ap * 2048 + bp];
      if (!v) continue;
      ++cnt; v -= 1;
      const int lane = v >> 8, r = v & 255;
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 31;
      if (a_row[ap] >= 0 && a_row[ap] != i) printf("A pos %d: inconsistent row\n", ap);
      a_row[ap] = i;
      if (b_col[bp] >= 0 && b_col[bp] != j) printf("B pos %d: inconsistent col\n", bp);
      b_col[bp] = j;
      if (a_kid[ap] < 0 && b_kid[bp] < 0) { a_kid[ap] = b_kid[bp] = nk++; }
      else if (a_kid[ap] < 0) a_kid[ap] = b_kid[bp];
      else if (b_kid[bp] < 0) b_kid[bp] = a_kid[ap];
      else if (a_kid[ap] != b_kid[bp]) printf("k class clash at A %d B %d\n", ap, bp);
    }
    if (cnt != 32) printf("A pos %d (lane %d byte %d): %d matches\n", ap, ap >> 5, ap & 31, cnt);
  }
  printf("k classes found: %d (expect 64)\n", nk);
  // hypothesis check: row = lane & 31 for A, col = lane & 31 for B; A k-class == B k-class at the same (lane >> 5, byte)
  int bad_row = 0, bad_same = 0;
  for (int p = 0; p < 2048; ++p) {
    if (a_row[p] != ((p >> 5) & 31)) ++bad_row;
    if (b_col[p] != ((p >> 5) & 31)) ++bad_row;
    // the B position with the same (lane >> 5, byte) in ANY lane of that half must share A's k class
    if (a_kid[p] != b_kid[p]) ++bad_same;
  }
  printf("row/col = lane & 31: %s;  same (lane>>5, byte) <=> same k in A and B: %s\n", bad_row ? "NO" : "yes", bad_same ? "NO" : "yes");
  // does k depend only on (lane >> 5, byte)?  print the class id table for lane 0 and lane 32, and check all lanes agree
  int dep = 0;
  for (int p = 0; p < 2048; ++p) { const int l = p >> 5, b = p & 31; if (a_kid[p] != a_kid[((l & 32)) * 32 + b]) ++dep; }
  printf("k depends only on (lane >> 5, byte): %s\n", dep ? "NO" : "yes");
  for (int hh = 0; hh < 2; ++hh) { printf("lane half %d: k class of bytes 0..31:", hh); for (int b = 0; b < 32; ++b) printf(" %d", a_kid[(32 * hh) * 32 + b]); printf("\n"); }

  // scale ownership
  float* ds; hipMalloc(&ds, 64 * 64 * 16 * 4);
  std::vector<float> hs(64 * 64 * 16);
  printf("scale ownership (A operand): for scale set in lane ls, which (lane-half of the data, byte) are doubled:\n");
  for (int lsx : {5, 37}) for (int dhalf = 0; dhalf < 2; ++dhalf) {
    printf("  scale x2 in lane %2d, data bytes 0..31 of lane %2d:", lsx, (lsx & 31) + 32 * dhalf);
    for (int which = 0; which < 32; ++which) {
      scale_owner<<<64, 64>>>(ds, which, dhalf);
      hipMemcpy(hs.data(), ds, hs.size() * 4, hipMemcpyDeviceToHost);
      // D[i][j] for this ls: rows where value == 2*(count) ... with only byte `which` of every lane set: row i gets 1 from lane i (k of half 0) + 1 from lane i+32
      // find row (lsx & 31): value 2 = none doubled, 3 = one doubled, 4 = both
      const int i = lsx & 31;
      // locate D[i][0]: lane = 0 + 32 * ((i >> 2) & 1), reg = (i & 3) + 4 * (i >> 3)
      const int lane = 32 * ((i >> 2) & 1), r = (i & 3) + 4 * (i >> 3);
      const float v = hs[((size_t)lsx * 64 + lane) * 16 + r];
      printf(" %g", v);
    }
    printf("\n");
  }
  return 0;
}
