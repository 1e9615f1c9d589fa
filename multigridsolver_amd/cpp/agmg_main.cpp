// agmg_main.cpp — drop-in for the reference's setup programs
//   src/CPU_C++/main.cpp:153-239   (./main <matrix_basename> <ktg> <npass> <tou> → <name>promatrix_cpu.mtx)
//   src/GPU_CUDAC++/main.cu:18-297 (same argv                                  → <name>promatrix_gpu.mtx)
// Reads ../../matrices/<name>.mtx, runs the multiple pairwise aggregation on the GPU (mgs_hier_coarsen,
// one level) and writes ../../matrices/<name>promatrix_gpu.mtx with writeMatrix semantics, so the
// reference's own `bicg <name> gpu` (src/common/bicg.cpp:150-151) consumes it unchanged.
#include "mgs_host.hpp"

using namespace mgs;

int main(int argc, char **argv) {
  if (argc != 5) {
    printf("Invalid arguments.\n");
    printf("Usage: %s <matrix_basename> <ktg> <npass> <tou>\n", argv[0]);
    printf("Example: %s mymatrix 10.0 2 4.0\n", argv[0]);
    exit(1);
  }
  try {
    std::string name = argv[1];
    const double ktg = std::stod(argv[2]); const int npass = std::stoi(argv[3]); const double tou = std::stod(argv[4]);
    const char *e = getenv("MGS_MATRIX_DIR");
    std::string dir = e ? std::string(e) + "/" : std::string("../../matrices/");
    std::cout << "Starting AGMG process with parameters:" << std::endl;
    std::cout << "  Matrix basename: " << name << std::endl << "  ktg: " << ktg << std::endl << "  npass: " << npass << std::endl << "  tou: " << tou << std::endl;
    SMatrix A = readMatrix(dir + name + ".mtx");
    std::cout << "Matrix A loaded: " << A.rows() << "x" << A.cols() << ", " << A.nonZeros() << " non-zeros." << std::endl;
    DeviceMatrix Ad(A);
    TicToc timer("AGMG Core Algorithm Time", 4);
    timer.tic();
    mgs_hier *h = nullptr;
    check(mgs_hier_create(context(), Ad.handle(), 0.6, 1, 1, &h), context());
    check(mgs_hier_coarsen(h, ktg, npass, tou, /*coarse_rows=*/0, /*max_levels=*/2), context());
    timer.toc();
    if (mgs_hier_nlev(h) < 2) { fprintf(stderr, "aggregation produced no coarse level\n"); return 1; }
    int nf = 0, nc = 0, isagg = 0;
    const mgs_xfer *T = mgs_hier_level_P(h, 0);
    check(mgs_xfer_shape(T, &nf, &nc, &isagg), context());
    std::vector<int> agg((size_t)nf);
    check(mgs_xfer_download_agg(T, agg.data()), context());
    SMatrix P; P.m_rows = nf; P.m_cols = nc; P.rowptr.assign(1, 0);
    for (int i = 0; i < nf; ++i) {                       // one unit entry per aggregated row (AGMG.cpp:181-186)
      if (agg[i] >= 0) { P.col.push_back(agg[i]); P.val.push_back(1.0); }
      P.rowptr.push_back((int)P.col.size());
    }
    printScreen(8, "AGMG completed, matrix size", nc);
    std::cout << "Prolongation matrix P created: " << P.rows() << "x" << P.cols() << ", " << P.nonZeros() << " non-zeros." << std::endl;
    std::string out = dir + name + "promatrix_gpu.mtx";
    std::cout << "Writing P matrix to: " << out << std::endl;
    writeMatrix(out, P);
    std::cout << "P matrix successfully written." << std::endl;
    mgs_hier_destroy(h);
  } catch (const std::exception &ex) {
    fprintf(stderr, "mgs_agmg: %s\n", ex.what());
    return 1;
  }
  return 0;
}
