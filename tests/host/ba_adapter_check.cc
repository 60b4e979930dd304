/*
 * ba_adapter_check.cc -- TEST INFRASTRUCTURE (built by tests/host/Makefile, run on the GPU
 * box by tests/test_ba_gpu.py).  Instantiates orthosfm_amd/host/ba_hip_adapter.h -- the C++
 * body of orthosfm::runBundleAdjustment -- with TEST DOUBLES of the reference classes: they
 * expose the accessors the reference's function uses (Camera.h:25-55,
 * OrthoQuaternionCamera.h:47-91, OrthographicCamera.h:66-134, track.h:21-107) and nothing
 * else, because the real headers need Eigen / OpenCV / Ceres, which this image lacks.
 * Reads a flattened scene from stdin (written by the test), runs the adapter and prints
 * cameras and points with 17 significant digits; the test compares them with the Python
 * mirror (orthosfm_amd/ba.py::run_bundle_adjustment) on the same scene: bit for bit.
 */
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "ba_hip_adapter.h"

struct View { unsigned id; int w, h; unsigned getID() const { return id; } int getWidth() const { return w; } int getHeight() const { return h; } };
struct Vec4 { double v[4]; Vec4() : v{0, 0, 0, 0} {} Vec4(double a, double b, double c, double d) : v{a, b, c, d} {}
              double* data() { return v; } const double* data() const { return v; } };
struct Feature { unsigned viewID, localFeatureID, globalFeatureID; float x, y; };
struct Track {
    std::vector<Feature> f; Vec4 p; bool has = false;
    void add(Feature x) { f.push_back(x); }
    unsigned size() const { return (unsigned)f.size(); }
    const Feature& get(int i) const { return f[i]; }
    const Vec4& getPoint() const { return p; } Vec4& getPoint() { return p; }
    void setPoint(Vec4 q) { p = q; has = true; }
    bool hasPoint() const { return has; }
};
struct Camera {
    std::shared_ptr<View> view; bool fixed = false;
    virtual ~Camera() {}
    const std::shared_ptr<View>& getView() const { return view; }
    const bool& isFixed() const { return fixed; }
};
struct QuatCam : Camera {
    double rot[4] = {0, 0, 0, 1}, ox = 0, oy = 0, sc = 1; bool fr = false, fo = false, fs = true;
    double* getRotation() { return rot; } double* getOffsetX() { return &ox; } double* getOffsetY() { return &oy; } double* getScale() { return &sc; }
    bool getRotationFixed() { return fr; } bool getOffsetFixed() { return fo; } bool getScaleFixed() { return fs; }
};
struct EulerCam : Camera {
    double phi = 0, theta = 0, roll = 0, ox = 0, oy = 0, sc = 1; int dof = 4;
    double* getPhi() { return &phi; } double* getTheta() { return &theta; } double* getRoll() { return &roll; }
    double* getOffsetX() { return &ox; } double* getOffsetY() { return &oy; } double* getScale() { return &sc; }
    bool getPhiFixed() const { return dof < 1; } bool getThetaFixed() const { return dof < 2; } bool getRollFixed() const { return dof < 3; }
    bool getOffsetFixed() const { return dof < 4; } bool getScaleFixed() const { return dof < 5; }
};
struct Algo { std::string name; std::string getName() { return name; } };

int main()
{
    int model, C, T, retri;
    if (scanf("%d %d %d %d", &model, &C, &T, &retri) != 4) return 2;
    std::vector<std::shared_ptr<Camera>> cams;
    for (int i = 0; i < C; ++i) {
        unsigned id; int w, h, fixed; double p[7];
        if (scanf("%u %d %d %d", &id, &w, &h, &fixed) != 4) return 2;
        for (double& x : p) if (scanf("%lf", &x) != 1) return 2;
        auto v = std::make_shared<View>(View{id, w, h});
        if (model == 0) { auto c = std::make_shared<QuatCam>(); c->view = v; c->fixed = fixed; for (int k = 0; k < 4; ++k) c->rot[k] = p[k]; c->ox = p[4]; c->oy = p[5]; c->sc = p[6]; cams.push_back(c); }
        else { auto c = std::make_shared<EulerCam>(); c->view = v; c->fixed = fixed; c->phi = p[0]; c->theta = p[1]; c->roll = p[2]; c->ox = p[3]; c->oy = p[4]; c->sc = p[5]; cams.push_back(c); }
    }
    std::vector<Track> tracks(T);
    for (int t = 0; t < T; ++t) {
        int n, has; double P[4];
        if (scanf("%d %d %lf %lf %lf %lf", &n, &has, &P[0], &P[1], &P[2], &P[3]) != 6) return 2;
        if (has) tracks[t].setPoint(Vec4(P[0], P[1], P[2], P[3]));
        for (int k = 0; k < n; ++k) { Feature f; if (scanf("%u %u %f %f", &f.viewID, &f.localFeatureID, &f.x, &f.y) != 4) return 2; f.globalFeatureID = 32768 * f.viewID + f.localFeatureID; tracks[t].add(f); }
    }
    auto algo = std::make_shared<Algo>();
    algo->name = model == 0 ? "Ortho Quaternion Reconstruction" : "Orthographic Reconstruction";
    try {
        osfm_adapter::runBundleAdjustment<QuatCam, EulerCam, Vec4>(cams, tracks, algo, true, retri != 0);
    } catch (std::exception const& e) { fprintf(stderr, "%s\n", e.what()); return 1; }
    for (int i = 0; i < C; ++i) {
        if (model == 0) { auto c = std::dynamic_pointer_cast<QuatCam>(cams[i]); printf("CAM %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", c->rot[0], c->rot[1], c->rot[2], c->rot[3], c->ox, c->oy, c->sc); }
        else { auto c = std::dynamic_pointer_cast<EulerCam>(cams[i]); printf("CAM %.17g %.17g %.17g %.17g %.17g %.17g 0\n", c->phi, c->theta, c->roll, c->ox, c->oy, c->sc); }
    }
    for (int t = 0; t < T; ++t) printf("PT %d %.17g %.17g %.17g %.17g\n", tracks[t].has ? 1 : 0, tracks[t].p.v[0], tracks[t].p.v[1], tracks[t].p.v[2], tracks[t].p.v[3]);
    return 0;
}
