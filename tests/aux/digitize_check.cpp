// Host check of the O(1) np.digitize(v, np.linspace(lo, hi, n)) used by the kernels (csrc/tg_kernels.hpp digitize_linspace; the two bodies are kept
// identical): against the edge-by-edge count, at and next to every edge of random bin sets.  Run by tests/test_host_cpu.py.
#include <cstdio>
#include <cmath>
#include <cstdlib>
static int ref(double v, double lo, double hi, int n) { const double step = (hi - lo) / (double)(n - 1); int c = 0; for (int k = 0; k < n; ++k) { const double prod = (double)k * step; const double e = (k == n - 1) ? hi : prod + lo; c += (e <= v) ? 1 : 0; } return c; }
static int fast(double v, double lo, double hi, int n) {
    const double step = (hi - lo) / (double)(n - 1);
    double t = floor((v - lo) / step);
    t = t < -1.0 ? -1.0 : (t > (double)n ? (double)n : t);
    int base = (int)t - 1; base = base < 0 ? 0 : (base > n ? n : base);
    int count = base;
    for (int j = 0; j < 4; ++j) { const int k = base + j; const double prod = (double)k * step; const double e = (k == n - 1) ? hi : prod + lo; count += (k < n && e <= v) ? 1 : 0; }
    return count;
}
int main() {
    unsigned long long st = 12345; long bad = 0, tot = 0;
    for (int trial = 0; trial < 400; ++trial) {
        st = st * 6364136223846793005ull + 1442695040888963407ull; double lo = -0.5 + (st >> 11) * (1.0 / 9007199254740992.0);
        st = st * 6364136223846793005ull + 1442695040888963407ull; double hi = lo + 0.01 + (st >> 11) * (1.0 / 9007199254740992.0);
        int n = 2 + (int)(st % 127);
        const double step = (hi - lo) / (n - 1);
        for (int k = -2; k < n + 2; ++k) for (int d = -3; d <= 3; ++d) {       // values at and next to every edge
            double e = (k == n - 1) ? hi : (double)k * step + lo; double v = e;
            for (int i = 0; i < (d < 0 ? -d : d); ++i) v = nextafter(v, d < 0 ? -1e300 : 1e300);
            ++tot; if (ref(v, lo, hi, n) != fast(v, lo, hi, n)) ++bad;
        }
        for (int i = 0; i < 2000; ++i) { st = st * 6364136223846793005ull + 1442695040888963407ull; double v = lo - 0.2 + (hi - lo + 0.4) * ((st >> 11) * (1.0 / 9007199254740992.0)); ++tot; if (ref(v, lo, hi, n) != fast(v, lo, hi, n)) ++bad; }
    }
    printf("%ld mismatches of %ld\n", bad, tot); return bad != 0;
}
