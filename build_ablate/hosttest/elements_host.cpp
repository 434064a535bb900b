#include "hip/hip_runtime.h"
#include "../../ssa-gym_amd/csrc/ssa_math.hpp"
#include <cstdio>
#include <vector>
namespace ssa { Vec6 kepler_general_v(Vec6 x, double tof) { Vec6 o; for (int i=0;i<6;++i) o.v[i]=NAN; return o; }
Vec8 kepler_general_diag_v(Vec6 x, double tof, Vec6* out) { Vec8 d; return d; } }
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); std::vector<double> x; double b[6]; while (fread(b, 8, 6, f) == 6) for (int i=0;i<6;++i) x.push_back(b[i]); fclose(f);
    int n = x.size()/6, bad = 0;
    FILE* o = fopen(argv[2], "wb");
    for (int j = 0; j < n; ++j) { double out[6]; bool ok = ssa::kepler_elements_fast(&x[6*j], 20.0, out); if (!ok || out[0] != out[0]) { if (bad < 5) printf("row %d ok %d out %g\n", j, (int)ok, out[0]); ++bad; } fwrite(out, 8, 6, o); }
    fclose(o); printf("n %d bad %d\n", n, bad); return 0; }
