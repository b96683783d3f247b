// ltc_fit.cpp — generates the two 64x64 RGBA32F tables the analytic (LTC) image needs: texSamplers[0] = LTC1 (the inverse
// transform, 4 coefficients), texSamplers[1] = LTC2 (BRDF magnitude, Fresnel weight, 0, clipped-sphere form factor).
//
// The reference ships these as a data header credited to learnopengl.com (external/LUT/ltc_matrix.h, uploaded at
// src/app/setup/create_scene.cppm:162-214, read at src/shaders/raygen.rgen:144-157 and LTC.glsl:61).  That header is third-party
// data and stays where it is; this program restates the PUBLISHED procedure that produces such tables — Heitz, Dupuy, Hill,
// Neubelt, "Real-Time Polygonal-Light Shading with Linearly Transformed Cosines", SIGGRAPH 2016, section 5 and its supplemental
// fitting notes — for the isotropic GGX microfacet BRDF (Smith height-correlated masking-shadowing):
//   for every roughness (64 values, alpha = roughness^2) and view elevation (64 values, cos(theta) = 1 - t^2):
//     1. the BRDF's norm, its Fresnel-weighted norm and its mean direction, by importance sampling (32 x 32 stratified samples);
//     2. an LTC D_o(M^-1 w) with M = frame(mean direction) * [m11 0 m13; 0 m22 0; 0 0 1], fitted by Nelder-Mead (downhill simplex)
//        to minimise the cubed absolute difference to the cosine-weighted BRDF, sampled from both distributions, warm-started
//        from the neighbouring entry;
//     3. stored: M^-1 scaled so its middle element is 1 (its elements 00, 02, 20, 22), and (norm, Fresnel norm);
//   plus the projected solid angle of a horizon-clipped spherical cap as a function of (elevation of its axis, sin^2 of its half
//   angle), in closed form (Snyder, "Area light sources for real-time graphics", 1996) — the .w channel of LTC2.
// The result is compared with the reference's header where that is present (tests/test_ltc_tables.py: the tables agree to the
// tolerance stated there; they cannot agree bit for bit, a downhill simplex is sensitive to its start and its arithmetic).
//
//   g++ -O2 -std=c++17 -pthread ltc_fit.cpp -o ltc_fit && ./ltc_fit ../../data/ltc_tables.bin
// Output: 2 x 64 x 64 x 4 little-endian float32 (LTC1 then LTC2), row-major with u = roughness fastest, v = sqrt(1 - cos(theta)).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr int kN = 64;          // table resolution
constexpr int kSamples = 32;    // stratified samples per dimension
constexpr double kMinAlpha = 1e-5;

struct V3 { double x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0 / length(a)); }

struct M3 {      // row-major
    double m[3][3];
    V3 mul(V3 v) const { return {m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z, m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z}; }
    double det() const {
        return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    }
    M3 inverse() const {
        const double d = 1.0 / det();
        M3 r;
        r.m[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) * d; r.m[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * d; r.m[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * d;
        r.m[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * d; r.m[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * d; r.m[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * d;
        r.m[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * d; r.m[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * d; r.m[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * d;
        return r;
    }
};
inline M3 matmul(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}

// ---- the BRDF being fitted: GGX, Smith height-correlated masking-shadowing; value includes the cosine of the light direction ----
struct Ggx {
    static double lambda(double alpha, double cosTheta) {
        if (cosTheta >= 1.0) return 0.0;
        const double a = 1.0 / alpha / std::tan(std::acos(cosTheta));
        return 0.5 * (-1.0 + std::sqrt(1.0 + 1.0 / a / a));
    }
    static double eval(V3 V, V3 L, double alpha, double& pdf) {
        if (V.z <= 0) { pdf = 0; return 0; }
        const double lambdaV = lambda(alpha, V.z);
        double G2 = 0;
        if (L.z > 0) G2 = 1.0 / (1.0 + lambdaV + lambda(alpha, L.z));
        const V3 H = normalize(V + L);
        const double sx = H.x / H.z, sy = H.y / H.z;
        double D = 1.0 / (1.0 + (sx * sx + sy * sy) / alpha / alpha);
        D = D * D / (kPi * alpha * alpha * H.z * H.z * H.z * H.z);
        pdf = std::fabs(D * H.z / 4.0 / dot(V, H));
        return D * G2 / 4.0 / V.z;
    }
    static V3 sample(V3 V, double alpha, double u1, double u2) {
        const double phi = 2.0 * kPi * u1, r = alpha * std::sqrt(u2 / (1.0 - u2));
        const V3 N = normalize(V3{r * std::cos(phi), r * std::sin(phi), 1.0});
        return N * (2.0 * dot(N, V)) - V;
    }
};

// ---- a linearly transformed clamped cosine --------------------------------------------------------------------------------
struct Ltc {
    double magnitude = 1, fresnel = 1, m11 = 1, m22 = 1, m13 = 0;
    V3 X{1, 0, 0}, Y{0, 1, 0}, Z{0, 0, 1};
    M3 M{}, invM{};
    double detM = 1;
    void update() {
        const M3 frame{{{X.x, Y.x, Z.x}, {X.y, Y.y, Z.y}, {X.z, Y.z, Z.z}}};       // columns X, Y, Z
        const M3 shape{{{m11, 0, m13}, {0, m22, 0}, {0, 0, 1}}};
        M = matmul(frame, shape);
        invM = M.inverse();
        detM = std::fabs(M.det());
    }
    double eval(V3 L) const {
        const V3 Lo = normalize(invM.mul(L));
        const V3 L_ = M.mul(Lo);
        const double l = length(L_), jacobian = detM / (l * l * l);
        const double D = std::max(0.0, Lo.z) / kPi;
        return magnitude * D / jacobian;
    }
    V3 sample(double u1, double u2) const {
        const double theta = std::acos(std::sqrt(u1)), phi = 2.0 * kPi * u2;
        return normalize(M.mul(V3{std::sin(theta) * std::cos(phi), std::sin(theta) * std::sin(phi), std::cos(theta)}));
    }
};

void average_terms(V3 V, double alpha, double& norm, double& fresnel, V3& averageDir) {
    norm = 0; fresnel = 0; averageDir = {0, 0, 0};
    for (int j = 0; j < kSamples; ++j)
        for (int i = 0; i < kSamples; ++i) {
            const double u1 = (i + 0.5) / kSamples, u2 = (j + 0.5) / kSamples;
            const V3 L = Ggx::sample(V, alpha, u1, u2);
            double pdf;
            const double value = Ggx::eval(V, L, alpha, pdf);
            if (pdf > 0) {
                const double w = value / pdf;
                const V3 H = normalize(V + L);
                norm += w;
                fresnel += w * std::pow(1.0 - std::max(dot(V, H), 0.0), 5.0);
                averageDir = averageDir + L * w;
            }
        }
    norm /= double(kSamples * kSamples);
    fresnel /= double(kSamples * kSamples);
    averageDir.y = 0;
    averageDir = normalize(averageDir);
}

double fit_error(const Ltc& ltc, V3 V, double alpha) {
    double error = 0;
    const auto term = [&](V3 L) {
        double pdfBrdf;
        const double b = Ggx::eval(V, L, alpha, pdfBrdf);
        const double l = ltc.eval(L), pdfLtc = l / ltc.magnitude;
        const double d = std::fabs(b - l);
        const double e = d * d * d / (pdfLtc + pdfBrdf);
        if (e == e && std::isfinite(e)) error += e;
    };
    for (int j = 0; j < kSamples; ++j)
        for (int i = 0; i < kSamples; ++i) {
            const double u1 = (i + 0.5) / kSamples, u2 = (j + 0.5) / kSamples;
            term(ltc.sample(u1, u2));                 // where the LTC puts its energy
            term(Ggx::sample(V, alpha, u1, u2));      // where the BRDF puts its energy
        }
    return error / double(kSamples * kSamples);
}

// Downhill simplex (Nelder & Mead 1965) in DIM dimensions: reflection 1, expansion 2, contraction 1/2, shrink 1/2.
template <int DIM>
double downhill_simplex(std::array<double, DIM>& best, const std::array<double, DIM>& start, double delta, double tolerance, int maxIters,
                        const std::function<double(const std::array<double, DIM>&)>& f) {
    using P = std::array<double, DIM>;
    std::array<P, DIM + 1> s;
    std::array<double, DIM + 1> fv;
    s[0] = start;
    for (int i = 1; i <= DIM; ++i) { s[i] = start; s[i][i - 1] += delta; }
    for (int i = 0; i <= DIM; ++i) fv[i] = f(s[i]);
    const auto along = [](const P& a, const P& b, double t) { P r; for (int k = 0; k < DIM; ++k) r[k] = a[k] + (b[k] - a[k]) * t; return r; };
    int lo = 0;
    for (int it = 0; it < maxIters; ++it) {
        lo = 0; int hi = 0, nh = 0;
        for (int i = 1; i <= DIM; ++i) { if (fv[i] < fv[lo]) lo = i; if (fv[i] > fv[hi]) hi = i; }
        nh = lo;
        for (int i = 0; i <= DIM; ++i) if (i != hi && fv[i] > fv[nh]) nh = i;
        const double a = std::fabs(fv[lo]), b = std::fabs(fv[hi]);
        if (2.0 * std::fabs(a - b) < (a + b) * tolerance) break;
        P centroid{};
        for (int i = 0; i <= DIM; ++i) if (i != hi) for (int k = 0; k < DIM; ++k) centroid[k] += s[i][k] / DIM;
        const P reflected = along(centroid, s[hi], -1.0);
        const double fr = f(reflected);
        if (fr < fv[lo]) {
            const P expanded = along(centroid, s[hi], -2.0);
            const double fe = f(expanded);
            if (fe < fr) { s[hi] = expanded; fv[hi] = fe; } else { s[hi] = reflected; fv[hi] = fr; }
        } else if (fr < fv[nh]) {
            s[hi] = reflected; fv[hi] = fr;
        } else {
            const P contracted = along(centroid, s[hi], 0.5);
            const double fc = f(contracted);
            if (fc < fv[hi]) { s[hi] = contracted; fv[hi] = fc; }
            else
                for (int i = 0; i <= DIM; ++i) if (i != lo) { s[i] = along(s[lo], s[i], 0.5); fv[i] = f(s[i]); }
        }
    }
    lo = 0;
    for (int i = 1; i <= DIM; ++i) if (fv[i] < fv[lo]) lo = i;
    best = s[lo];
    return fv[lo];
}

struct Entry { M3 M; double magnitude, fresnel; double m11, m22, m13; };

// one roughness column, view elevations from normal incidence to grazing, each warm-started from the one before
void fit_column(int a, const Entry* isoStart, std::vector<Entry>& tab) {
    const double roughness = a / double(kN - 1);
    const double alpha = std::max(roughness * roughness, kMinAlpha);
    for (int t = 0; t < kN; ++t) {
        const double x = t / double(kN - 1);
        const double ct = 1.0 - x * x;
        const double theta = std::min(1.57, std::acos(ct));
        const V3 V{std::sin(theta), 0, std::cos(theta)};
        Ltc ltc;
        V3 averageDir;
        average_terms(V, alpha, ltc.magnitude, ltc.fresnel, averageDir);
        const bool isotropic = t == 0;
        std::array<double, 3> start{};
        if (isotropic) {
            ltc.X = {1, 0, 0}; ltc.Y = {0, 1, 0}; ltc.Z = {0, 0, 1};
            start = isoStart ? std::array<double, 3>{isoStart->m11, isoStart->m22, 0.0} : std::array<double, 3>{1.0, 1.0, 0.0};
        } else {
            const V3 L = averageDir;
            ltc.X = {L.z, 0, -L.x}; ltc.Y = {0, 1, 0}; ltc.Z = L;
            const Entry& prev = tab[(size_t)a + (size_t)(t - 1) * kN];
            start = {prev.m11, prev.m22, prev.m13};
        }
        const auto apply = [&](double p0, double p1, double p2) {
            ltc.m11 = std::max(p0, 1e-7); ltc.m22 = isotropic ? ltc.m11 : std::max(p1, 1e-7); ltc.m13 = isotropic ? 0.0 : p2;
            ltc.update();
        };
        if (isotropic) {
            std::array<double, 1> best{}, s0{start[0]};
            downhill_simplex<1>(best, s0, 0.05, 1e-5, 100, [&](const std::array<double, 1>& p) { apply(p[0], p[0], 0); return fit_error(ltc, V, alpha); });
            apply(best[0], best[0], 0);
        } else {
            std::array<double, 3> best{};
            downhill_simplex<3>(best, start, 0.05, 1e-5, 100, [&](const std::array<double, 3>& p) { apply(p[0], p[1], p[2]); return fit_error(ltc, V, alpha); });
            apply(best[0], best[1], best[2]);
        }
        Entry e;
        e.M = ltc.M; e.magnitude = ltc.magnitude; e.fresnel = ltc.fresnel; e.m11 = ltc.m11; e.m22 = ltc.m22; e.m13 = ltc.m13;
        e.M.m[0][1] = 0; e.M.m[1][0] = 0; e.M.m[2][1] = 0; e.M.m[1][2] = 0;      // zero by symmetry: remove the numerical dust
        tab[(size_t)a + (size_t)t * kN] = e;
    }
}

// ---- projected solid angle of a spherical cap clipped by the horizon (closed form) ---------------------------------------------
double sqr(double x) { return x * x; }
double cap_g(double w, double s, double g) { return -2.0 * std::sin(w) * std::cos(s) * std::cos(g) + kPi / 2.0 - g + std::sin(g) * std::cos(g); }
double cap_h(double w, double s, double g) {
    const double sinsSq = sqr(std::sin(s)), cosgSq = sqr(std::cos(g));
    return std::cos(w) * (std::cos(g) * std::sqrt(std::max(sinsSq - cosgSq, 0.0)) + sinsSq * std::asin(std::min(1.0, std::cos(g) / std::sin(s))));
}
double clipped_cap(double w, double s) {       // w: elevation of the axis from the normal, s: half angle
    const double sinsSq = sqr(std::sin(s));
    if (w >= 0.0 && w <= kPi / 2.0 - s) return kPi * std::cos(w) * sinsSq;
    const double g = std::asin(std::max(-1.0, std::min(1.0, std::cos(s) / std::sin(w))));
    if (w >= kPi / 2.0 - s && w < kPi / 2.0) return kPi * std::cos(w) * sinsSq + cap_g(w, s, g) - cap_h(w, s, g);
    if (w >= kPi / 2.0 && w < kPi / 2.0 + s) return cap_g(w, s, g) + cap_h(w, s, g);
    return 0.0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s out.bin [threads]\n", argv[0]); return 2; }
    const int threads = argc > 2 ? std::max(1, atoi(argv[2])) : 8;
    std::vector<Entry> tab((size_t)kN * kN);
    // the normal-incidence entries (t = 0) start from their rougher neighbour: rough to smooth, one after the other ...
    // (fit_column does t = 0 first; to keep columns independent the t = 0 chain is run first on its own)
    {
        std::vector<Entry> iso((size_t)kN);
        for (int a = kN - 1; a >= 0; --a) {
            std::vector<Entry> col((size_t)kN * kN);
            // only t = 0 of this column is needed here: fit it with a one-entry loop
            const double roughness = a / double(kN - 1), alpha = std::max(roughness * roughness, kMinAlpha);
            const V3 V{0, 0, 1};
            Ltc ltc; V3 avg;
            average_terms(V, alpha, ltc.magnitude, ltc.fresnel, avg);
            const double s0 = a == kN - 1 ? 1.0 : iso[(size_t)a + 1].m11;
            std::array<double, 1> best{}, st{s0};
            downhill_simplex<1>(best, st, 0.05, 1e-5, 100, [&](const std::array<double, 1>& p) {
                ltc.m11 = ltc.m22 = std::max(p[0], 1e-7); ltc.m13 = 0; ltc.update(); return fit_error(ltc, V, alpha); });
            iso[(size_t)a].m11 = iso[(size_t)a].m22 = std::max(best[0], 1e-7); iso[(size_t)a].m13 = 0;
        }
        // ... then every column on its own thread, its t = 0 entry re-fitted from the chain's value
        std::vector<std::thread> pool;
        for (int w = 0; w < threads; ++w)
            pool.emplace_back([&, w] { for (int a = w; a < kN; a += threads) fit_column(a, &iso[(size_t)a], tab); });
        for (auto& th : pool) th.join();
    }
    std::vector<float> out((size_t)2 * kN * kN * 4);
    for (int t = 0; t < kN; ++t)
        for (int a = 0; a < kN; ++a) {
            const Entry& e = tab[(size_t)a + (size_t)t * kN];
            M3 inv = e.M.inverse();
            const double s = 1.0 / inv.m[1][1];
            float* t1 = &out[((size_t)t * kN + a) * 4];
            // the shader rebuilds Minv = mat3(vec3(t1.x, 0, t1.y), vec3(0, 1, 0), vec3(t1.z, 0, t1.w)) (columns): x = row 0 col 0,
            // y = row 2 col 0, z = row 0 col 2, w = row 2 col 2
            t1[0] = (float)(inv.m[0][0] * s); t1[1] = (float)(inv.m[2][0] * s); t1[2] = (float)(inv.m[0][2] * s); t1[3] = (float)(inv.m[2][2] * s);
            float* t2 = &out[(size_t)kN * kN * 4 + ((size_t)t * kN + a) * 4];
            // sphere table: u = elevation (z = 2 u - 1), v = sin^2 of the half angle
            const double u1 = a / double(kN - 1), u2 = t / double(kN - 1);
            const double z = 2.0 * u1 - 1.0, len = u2;
            const double sigma = std::asin(std::sqrt(len)), omega = std::acos(z);
            const double sphere = sigma > 0.0 ? clipped_cap(omega, sigma) / (kPi * len) : std::max(z, 0.0);
            t2[0] = (float)e.magnitude; t2[1] = (float)e.fresnel; t2[2] = 0.0f; t2[3] = (float)sphere;
        }
    FILE* f = fopen(argv[1], "wb");
    if (!f) { perror(argv[1]); return 1; }
    fwrite(out.data(), sizeof(float), out.size(), f);
    fclose(f);
    double s1 = 0, s2 = 0;
    for (size_t i = 0; i < (size_t)kN * kN * 4; ++i) { s1 += out[i]; s2 += out[(size_t)kN * kN * 4 + i]; }
    printf("wrote %s: sum(LTC1) = %.6f, sum(LTC2) = %.6f\n", argv[1], s1, s2);
    return 0;
}
