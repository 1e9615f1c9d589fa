// bicg_main.cpp — drop-in for the reference solve driver src/common/bicg.cpp:138-180:
//   ./mgs_bicg <matrix_name> <cpu|gpu|device>
// reads ../../matrices/<name>.mtx and ../../matrices/<name>promatrix_<tag>.mtx (bicg.cpp:150-151;
// tag "device" = no P file: the whole hierarchy is aggregated on the GPU), builds the
// preconditioner, fills b with srand(0)/rand() (bicg.cpp:139,159-162), runs BiCGSTABiml with
// tol 1e-6 and max_iter 10000 (bicg.cpp:148,164) and prints the reference's two [info] lines.
// Environment knobs (new; the reference has no smoother options): MGS_OMEGA, MGS_NU1, MGS_NU2, MGS_SIGMA (over-correction),
// MGS_ADDITIVE=1 / MGS_NO_PRECOND=1 (the reference's two dead solve() switches, bicg.cpp:42-43,53-59),
// MGS_TOL, MGS_MATRIX_DIR, MGS_GENERIC=1 (run the generic operator-overloading BiCGSTABiml
// template instead of the fused device path).
#include "mgs_host.hpp"

using namespace mgs;
using std::string;

int main(int argc, char **argv) {
  srand(0);
  if (argc != 3) {
    printf("Incorrect number of arguments.\n");
    printf("Usage: $ ./mgs_bicg <matrix_name> <cpu|gpu|device>\n matrix_name is the name of the matrix in the matrices folder\n cpu/gpu selects <matrix_name>promatrix_<cpu|gpu>.mtx written by the AGMG setup; 'device' aggregates on the GPU\n");
    exit(1);
  }
  try {
    string matrix_name = argv[1], device = argv[2];
    const char *e;
    string dir = (e = getenv("MGS_MATRIX_DIR")) ? string(e) + "/" : string("../../matrices/");
    double tol = (e = getenv("MGS_TOL")) ? atof(e) : 1e-6;
    MultiGridPrecond::Options opt;
    if ((e = getenv("MGS_OMEGA"))) opt.omega = atof(e);
    if ((e = getenv("MGS_NU1"))) opt.nu1 = atoi(e);
    if ((e = getenv("MGS_NU2"))) opt.nu2 = atoi(e);
    if ((e = getenv("MGS_SIGMA"))) opt.correction_scale = atof(e);
    if ((e = getenv("MGS_ADDITIVE"))) opt.multiplicative_precond = atoi(e) == 0;      // the reference's two dead switches (bicg.cpp:42-43)
    if ((e = getenv("MGS_NO_PRECOND"))) opt.use_preconditioner = atoi(e) == 0;

    SMatrix A = readMatrix(dir + matrix_name + string(".mtx"));
    DeviceMatrix Ad(A);
    std::unique_ptr<MultiGridPrecond> precond;
    if (device == "device") precond.reset(new MultiGridPrecond(Ad, nullptr, opt));
    else {
      SMatrix P_matrix = readMatrix(dir + matrix_name + string("promatrix_") + device + string(".mtx"));
      precond.reset(new MultiGridPrecond(Ad, &P_matrix, opt));
    }

    VectorXd x(A.rows());
    x.setZero();
    srand(0);   // the reference draws nothing between srand(0) (:139) and this loop; runtime start-up here might
    VectorXd b(A.rows());
    for (int i = 0; i < A.rows(); i++) {       // bicg.cpp:159-162, statement for statement (host-staged element access)
      b[i] = rand() / (RAND_MAX + 0.0);
    }

    int max_iter = 10000;
    TicToc solverTimer("BiCGStab_SolveTimer", 4);
    solverTimer.tic();
    int status = getenv("MGS_GENERIC") ? BiCGSTABiml<DeviceMatrix, VectorXd, MultiGridPrecond, double>(Ad, x, b, *precond, max_iter, tol)
                                       : BiCGSTABiml(A, x, b, *precond, max_iter, tol);     // bicg.cpp:168 as written (A = the host SMatrix)
    solverTimer.toc();

    if (status == 0) {
      printScreen(4, "Tolerance ", tol);
      printScreen(4, "Number of iterations BICG", max_iter);
    } else {
      std::cout << "BiCGSTABiml encountered a problem with status code: " << status << std::endl;
    }
    if ((e = getenv("MGS_DUMP_X"))) {   // test hook: solution as raw little-endian f64
      std::vector<double> xh = x.download();
      FILE *f = fopen(e, "wb"); fwrite(xh.data(), 8, xh.size(), f); fclose(f);
    }
  } catch (const std::exception &ex) {
    fprintf(stderr, "mgs_bicg: %s\n", ex.what());
    return 1;
  }
  return 0;
}
