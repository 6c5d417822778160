// Skeleton of a CELL-CENTRIC linear-Poisson assembly pass ("measure it"): what the cheapest
// possible version of that design costs on gfx950, before any of its real problems (tile lists of unstructured meshes,
// record traffic, partial SELL lines, boundary conditions, a right-hand side) are paid for.
//
// Design being priced: rows are grouped into 3-D tiles of 16 x 8 x 8 = 1024 vertices (one workgroup of 1024 threads, the
// off-diagonal strips of its rows -- 14 slots x 1024 rows x 8 B = 112 KB -- and the coordinates of the 17 x 9 x 9
// vertices it touches in LDS).  Every tetrahedron that touches the tile (17 * 9 * 9 ... cells of the halo layer included:
// 16*8*8*6 interior-ish + halo = (17*9*9 - ...) is enumerated exactly below) is evaluated ONCE per tile: its six pair
// terms K_ab from the Gram matrix of three edge vectors, each added to the strip of both rows when they are in the tile
// (ds_add_f64, lane-random addresses).  Owner-computes (k_poisson_system_pipe) evaluates 24 visits x 3 pairs per row.
// The tile's strips are then written out coalesced (1.13 GB at the benchmark size, = the SELL values the real pass writes).
// NOT modelled (all of it extra cost): per-tile cell lists from HBM (here ONE list, L2-resident, shared by all tiles),
// 1/(36|T|) per cell, the diagonal, Dirichlet masks, the Newton right-hand side, ragged tiles, partial 128-B lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <algorithm>
#include <array>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int TX = 16, TY = 8, TZ = 8, ROWS = TX * TY * TZ;          // 1024 rows per tile
constexpr int HX = TX + 1, HY = TY + 1, HZ = TZ + 1;                  // vertices a tile's cells touch on the +side ... (one layer each side below)
constexpr int NB = 14;

struct CellRec32 {
  uint16_t vert[4];         // local vertex id in the (TX+2)(TY+2)(TZ+2) halo box
  int16_t row[4];           // local row (0..1023) or -1 when the vertex belongs to another tile
  uint8_t slot[12];         // slot of b in a's row for the 12 ordered pairs (a, b), a != b, in the order (0,1)(0,2)(0,3)(1,0)...
  uint32_t pad;
};

template <bool ATOMIC>
__global__ __launch_bounds__(1024) void k_cell_tile(int n_tiles, int n_cells, const CellRec32* __restrict__ rec,
                                                    const double* __restrict__ xhalo, int n_halo, double* __restrict__ vals) {
  extern __shared__ double lds[];
  double* strip = lds;                         // [NB][ROWS]
  double* px = lds + NB * ROWS;                // [3][n_halo]
  const int tid = threadIdx.x;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#pragma unroll
    for (int k = 0; k < NB; ++k) strip[k * ROWS + tid] = 0.0;
    for (int i = tid; i < 3 * n_halo; i += 1024) px[i] = xhalo[i] + 1e-9 * tile;      // the tile's coordinates (here: one table + a shift)
    __syncthreads();
    for (int c = tid; c < n_cells; c += 1024) {
      const CellRec32 r = rec[c];
      double p[4][3];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int d = 0; d < 3; ++d) p[a][d] = px[d * n_halo + r.vert[a]];
      // Gram matrix of the edges from vertex 0, then the six pair terms (P1 stiffness up to the factor 1/(36|T|))
      double e[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int d = 0; d < 3; ++d) e[i][d] = p[i + 1][d] - p[0][d];
      double g[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) { g[i][j] = e[i][0] * e[j][0] + e[i][1] * e[j][1] + e[i][2] * e[j][2]; g[j][i] = g[i][j]; }
      // cofactors of the Gram matrix: grad(lambda_i).grad(lambda_j) = cof_ij / det, i, j = 1..3; vertex 0 by row sums
      double cof[3][3];
      cof[0][0] = g[1][1] * g[2][2] - g[1][2] * g[1][2];
      cof[1][1] = g[0][0] * g[2][2] - g[0][2] * g[0][2];
      cof[2][2] = g[0][0] * g[1][1] - g[0][1] * g[0][1];
      cof[0][1] = g[0][2] * g[1][2] - g[0][1] * g[2][2];
      cof[0][2] = g[0][1] * g[1][2] - g[0][2] * g[1][1];
      cof[1][2] = g[0][1] * g[0][2] - g[0][0] * g[1][2];
      const double det = g[0][0] * cof[0][0] + g[0][1] * cof[0][1] + g[0][2] * cof[0][2];
      const double w = 1.0 / (6.0 * sqrt(fabs(det)) + 1e-300);         // |T| = sqrt(det)/6; K_ij = |T| cof_ij / det = cof_ij / (6 sqrt(det))
      double K[4][4];
      K[1][2] = cof[0][1] * w; K[1][3] = cof[0][2] * w; K[2][3] = cof[1][2] * w;
      K[0][1] = -(cof[0][0] + cof[0][1] + cof[0][2]) * w;
      K[0][2] = -(cof[0][1] + cof[1][1] + cof[1][2]) * w;
      K[0][3] = -(cof[0][2] + cof[1][2] + cof[2][2]) * w;
      int q = 0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (a == b) continue;
          const double kab = a < b ? K[a][b] : K[b][a];
          if (r.row[a] >= 0) {
            double* dst = &strip[r.slot[q] * ROWS + r.row[a]];
            if (ATOMIC) (void)__hip_atomic_fetch_add(dst, kab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else *dst += kab;                                         // (racy: only to price the atomics)
          }
          ++q;
        }
    }
    __syncthreads();
    double* out = vals + (size_t)tile * NB * ROWS;
#pragma unroll
    for (int k = 0; k < NB; ++k) out[k * ROWS + tid] = strip[k * ROWS + tid];
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 215;
  const long long n_rows = (long long)(n + 1) * (n + 1) * (n + 1);
  const int n_tiles = (int)((n_rows + ROWS - 1) / ROWS);
  // the halo box of a tile: one layer of vertices on every side
  const int BX = TX + 2, BY = TY + 2, BZ = TZ + 2, n_halo = BX * BY * BZ;
  auto vid = [&](int i, int j, int k) { return (k * BY + j) * BX + i; };                    // box coordinates 0..T+1
  auto row_of = [&](int i, int j, int k) -> int { return (i >= 1 && i <= TX && j >= 1 && j <= TY && k >= 1 && k <= TZ) ? ((k - 1) * TY + (j - 1)) * TX + (i - 1) : -1; };
  // Kuhn split of every cube of the box: 6 tets = the 6 monotone paths from corner (0,0,0) to (1,1,1)
  const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  std::vector<CellRec32> recs;
  // slot of column b in row a: index of the offset (b - a) among the 14 neighbours of the Kuhn mesh
  std::vector<std::array<int, 3>> nb_off;
  for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
    if (!dx && !dy && !dz) continue;
    const bool pos = dx >= 0 && dy >= 0 && dz >= 0, neg = dx <= 0 && dy <= 0 && dz <= 0;
    if (pos || neg) nb_off.push_back({dx, dy, dz});
  }
  if ((int)nb_off.size() != NB) { printf("neighbour count %zu\n", nb_off.size()); return 1; }
  auto slot_of = [&](int dx, int dy, int dz) { for (int s = 0; s < NB; ++s) if (nb_off[s][0] == dx && nb_off[s][1] == dy && nb_off[s][2] == dz) return s; return 0; };
  for (int k = 0; k < BZ - 1; ++k) for (int j = 0; j < BY - 1; ++j) for (int i = 0; i < BX - 1; ++i)
    for (int t = 0; t < 6; ++t) {
      int c[4][3] = {{i, j, k}, {i, j, k}, {i, j, k}, {i, j, k}};
      for (int s = 0; s < 3; ++s) for (int a = s + 1; a < 4; ++a) c[a][perm[t][s]] += 1;
      CellRec32 r{};
      bool any = false;
      for (int a = 0; a < 4; ++a) { r.vert[a] = (uint16_t)vid(c[a][0], c[a][1], c[a][2]); r.row[a] = (int16_t)row_of(c[a][0], c[a][1], c[a][2]); any |= r.row[a] >= 0; }
      if (!any) continue;
      int q = 0;
      for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) { if (a == b) continue; r.slot[q++] = (uint8_t)slot_of(c[b][0] - c[a][0], c[b][1] - c[a][1], c[b][2] - c[a][2]); }
      recs.push_back(r);
    }
  const int n_cells = (int)recs.size();
  std::vector<double> xh(3 * (size_t)n_halo);
  for (int k = 0; k < BZ; ++k) for (int j = 0; j < BY; ++j) for (int i = 0; i < BX; ++i) {
    const int v = vid(i, j, k);
    xh[v] = i * 0.01 + 1e-4 * ((v * 7) % 5); xh[n_halo + v] = j * 0.01 + 1e-4 * ((v * 3) % 7); xh[2 * n_halo + v] = k * 0.01 + 1e-4 * ((v * 5) % 3);
  }
  printf("n = %d: %lld rows, %d tiles of %d rows; %d cells per tile (%.2f per row; owner-computes visits 24 per row), %d pair evaluations per row (owner-computes: 72)\n",
         n, n_rows, n_tiles, ROWS, n_cells, (double)n_cells / ROWS, 6 * n_cells / ROWS);
  CellRec32* d_rec; double* d_x; double* d_vals;
  CHECK(hipMalloc(&d_rec, recs.size() * sizeof(CellRec32)));
  CHECK(hipMalloc(&d_x, xh.size() * sizeof(double)));
  CHECK(hipMalloc(&d_vals, (size_t)n_tiles * NB * ROWS * sizeof(double)));
  CHECK(hipMemcpy(d_rec, recs.data(), recs.size() * sizeof(CellRec32), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_x, xh.data(), xh.size() * sizeof(double), hipMemcpyHostToDevice));
  const size_t lds = ((size_t)NB * ROWS + 3 * (size_t)n_halo) * sizeof(double);
  CHECK(hipFuncSetAttribute((const void*)k_cell_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHECK(hipFuncSetAttribute((const void*)k_cell_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int variant = 0; variant < 2; ++variant) {
    for (int grid : {256, 512, n_tiles}) {
      float best = 1e30f;
      for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0));
        if (variant == 0) hipLaunchKernelGGL(k_cell_tile<true>, dim3(grid), dim3(1024), lds, 0, n_tiles, n_cells, d_rec, d_x, n_halo, d_vals);
        else hipLaunchKernelGGL(k_cell_tile<false>, dim3(grid), dim3(1024), lds, 0, n_tiles, n_cells, d_rec, d_x, n_halo, d_vals);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = std::min(best, ms);
      }
      printf("%s grid %6d: %.3f ms  (LDS %zu KB per workgroup; writes %.2f GB)\n", variant == 0 ? "ds_add_f64        " : "plain read-add-write",
             grid, best, lds / 1024, (double)n_tiles * NB * ROWS * 8 / 1e9);
    }
  }
  double chk = 0; std::vector<double> h(NB * ROWS);
  CHECK(hipMemcpy(h.data(), d_vals, h.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (double v : h) chk += v;
  printf("checksum of tile 0 (row sums of the off-diagonals): %.6e\n", chk);
  return 0;
}
