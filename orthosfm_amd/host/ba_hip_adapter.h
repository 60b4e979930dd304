// Drop-in body of orthosfm::runBundleAdjustment
// (src/bundle_adjustment/bundle_adjustment.h:18-20, bundle_adjustment.cpp:49-161) on top
// of the C ABI of include/osfm_hip.h.
//
// The reference's own types (Camera, OrthoQuaternionCamera, OrthographicCamera, Track,
// Feature, ReconstructionAlgorithm) pull in Eigen, OpenCV and Ceres, none of which is in
// the build image, so this header names them only through template parameters: inside
// the OrthoSfM tree it is instantiated with the real classes (INTEGRATION.md section 2
// shows the three lines), here with test doubles that expose the same accessors
// (tests/host/ba_adapter_check.cc, run on the GPU against the Python mirror).  It uses
// exactly the members the reference function uses: Camera::getView()->getID() /
// getWidth() / getHeight() / isFixed(), the raw parameter pointers and fixed flags of the
// two camera classes, Track::size() / get(i) / hasPoint() / getPoint() / setPoint() /
// add(), ReconstructionAlgorithm::getName().
#ifndef OSFM_BA_HIP_ADAPTER_HEADER
#define OSFM_BA_HIP_ADAPTER_HEADER

#include <cstdint>
#include <iostream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "osfm_hip.h"

namespace osfm_adapter {

// filterTracksToAvailableCameras(cameras, tracks, false, false) (src/util/common.cpp:85-139):
// every track restricted to the features whose view has a camera, kept when more than one
// is left; the copies carry NO point (a default-constructed Track).
template <class Track, class CameraPtr>
std::vector<Track> filter_tracks_to_available_cameras(std::vector<CameraPtr> const& cameras,
    std::vector<Track> const& tracks)
{
    std::map<unsigned int, int> have;
    for (auto const& c : cameras) have.insert({c->getView()->getID(), 1});
    std::vector<Track> out;
    for (auto const& t : tracks) {
        Track cur;
        for (int f = 0; f < (int)t.size(); ++f)
            if (have.count(t.get(f).viewID)) cur.add(t.get(f));
        if (cur.size() > 1) out.push_back(cur);
    }
    return out;
}

// QuatCamera / EulerCamera: the two concrete camera classes; Vec4: Eigen::Vector4d (anything
// constructible from four doubles that setPoint accepts).
template <class QuatCamera, class EulerCamera, class Vec4, class CameraPtr, class Track, class AlgorithmPtr>
void runBundleAdjustment(std::vector<CameraPtr>& cameras, std::vector<Track>& tracks,
    AlgorithmPtr const& algorithm, bool optimizePoints, bool retriangulatePoints, int device = 0)
{
    const bool quat = algorithm->getName() == "Ortho Quaternion Reconstruction";
    std::map<unsigned int, int> viewToCam;                                       // :54-57 (insert keeps the first)
    for (int i = 0; i < (int)cameras.size(); ++i)
        viewToCam.insert({cameras[i]->getView()->getID(), i});

    // :71-83: the filtered copy is what gets optimised (and thrown away) when the points are
    // re-triangulated; the triangulation itself runs on the device in front of the solve
    std::vector<Track>* work = &tracks;
    std::vector<Track> local;
    if (retriangulatePoints) {
        local = filter_tracks_to_available_cameras(cameras, tracks);
        work = &local;
    }

    const size_t C = cameras.size();
    std::vector<double> cam(7 * C, 0.0), pts, xy;
    std::vector<uint8_t> cst(7 * C, 0);
    std::vector<int32_t> w(C), h(C), oc, op, owner;
    for (size_t i = 0; i < C; ++i) {
        w[i] = cameras[i]->getView()->getWidth();
        h[i] = cameras[i]->getView()->getHeight();
        const bool fixed = cameras[i]->isFixed();
        if (quat) {                                                              // OrthoQuaternionRecoAlgorithm.cpp:121-148
            auto c = std::dynamic_pointer_cast<QuatCamera>(cameras[i]);
            if (!c) throw std::runtime_error("osfm: camera is not an OrthoQuaternionCamera");
            for (int k = 0; k < 4; ++k) cam[7 * i + k] = c->getRotation()[k];    // Eigen coeff order x, y, z, w
            cam[7 * i + 4] = *c->getOffsetX(); cam[7 * i + 5] = *c->getOffsetY(); cam[7 * i + 6] = *c->getScale();
            for (int k = 0; k < 4; ++k) cst[7 * i + k] = fixed || c->getRotationFixed();
            cst[7 * i + 4] = cst[7 * i + 5] = fixed || c->getOffsetFixed();
            cst[7 * i + 6] = fixed || c->getScaleFixed();
        } else {                                                                 // OrthographicReconstructionAlgorithm.cpp:148-178
            auto c = std::dynamic_pointer_cast<EulerCamera>(cameras[i]);
            if (!c) throw std::runtime_error("osfm: camera is not an OrthographicCamera");
            double* v[6] = {c->getPhi(), c->getTheta(), c->getRoll(), c->getOffsetX(), c->getOffsetY(), c->getScale()};
            const bool fx[6] = {c->getPhiFixed(), c->getThetaFixed(), c->getRollFixed(),
                                c->getOffsetFixed(), c->getOffsetFixed(), c->getScaleFixed()};
            for (int k = 0; k < 6; ++k) { cam[7 * i + k] = *v[k]; cst[7 * i + k] = fixed || fx[k]; }
            cst[7 * i + 6] = 1;
        }
    }
    for (int t = 0; t < (int)work->size(); ++t) {                                // :86-123
        if (!retriangulatePoints && !(*work)[t].hasPoint()) continue;
        const int j = (int)owner.size();
        owner.push_back(t);
        if ((*work)[t].hasPoint()) { const double* P = (*work)[t].getPoint().data(); pts.insert(pts.end(), P, P + 4); }
        else { const double P[4] = {0.0, 0.0, 0.0, 1.0}; pts.insert(pts.end(), P, P + 4); }
        for (int f = 0; f < (int)(*work)[t].size(); ++f) {
            auto it = viewToCam.find((*work)[t].get(f).viewID);
            if (it == viewToCam.end()) continue;
            xy.push_back((*work)[t].get(f).x); xy.push_back((*work)[t].get(f).y);   // float -> double (track.h:26-27)
            oc.push_back(it->second); op.push_back(j);
        }
    }
    std::vector<double> before = pts;
    osfm_ba_problem p;
    p.model = quat ? OSFM_BA_MODEL_QUATERNION : OSFM_BA_MODEL_EULER;
    p.num_cameras = (int32_t)C; p.num_points = (int32_t)owner.size(); p.num_observations = (int32_t)oc.size();
    p.cam_params = cam.data(); p.cam_const = cst.data(); p.img_width = w.data(); p.img_height = h.data();
    p.points = pts.data(); p.obs_xy = xy.data(); p.obs_camera = oc.data(); p.obs_point = op.data();
    if (osfm_version() != OSFM_ABI_VERSION)      // osfm_ba_summary is written in full by the library
        throw std::runtime_error("osfm: libosfm_hip.so has ABI version " + std::to_string(osfm_version()) +
            ", this adapter was built against " + std::to_string(OSFM_ABI_VERSION));
    osfm_ba_options o;
    osfm_ba_options_default(&o);
    o.optimize_points = optimizePoints ? 1 : 0;
    o.retriangulate_points = retriangulatePoints ? 1 : 0;
    o.device = device;
    osfm_ba_summary s;
    if (osfm_ba_solve(&p, &o, &s) != OSFM_OK)
        throw std::runtime_error(std::string("osfm: ") + osfm_last_error());

    // scatter back through the same raw pointers Ceres writes through
    for (size_t i = 0; i < C; ++i) {
        if (quat) {
            auto c = std::dynamic_pointer_cast<QuatCamera>(cameras[i]);
            for (int k = 0; k < 4; ++k) c->getRotation()[k] = cam[7 * i + k];
            *c->getOffsetX() = cam[7 * i + 4]; *c->getOffsetY() = cam[7 * i + 5]; *c->getScale() = cam[7 * i + 6];
        } else {
            auto c = std::dynamic_pointer_cast<EulerCamera>(cameras[i]);
            *c->getPhi() = cam[7 * i + 0]; *c->getTheta() = cam[7 * i + 1]; *c->getRoll() = cam[7 * i + 2];
            *c->getOffsetX() = cam[7 * i + 3]; *c->getOffsetY() = cam[7 * i + 4]; *c->getScale() = cam[7 * i + 5];
        }
    }
    for (size_t j = 0; j < owner.size(); ++j)
        (*work)[owner[j]].setPoint(Vec4(pts[4 * j], pts[4 * j + 1], pts[4 * j + 2], pts[4 * j + 3]));

    // summary.BriefReport() and the point-motion line (:148-160)
    static const char* kTerm[] = {"?", "CONVERGENCE", "CONVERGENCE", "CONVERGENCE", "CONVERGENCE", "NO_CONVERGENCE", "FAILURE"};
    std::cout << "Ceres Solver Report: Iterations: " << s.num_iterations + 1 << ", Initial cost: " << s.initial_cost
              << ", Final cost: " << s.final_cost << ", Termination: " << kTerm[s.termination >= 0 && s.termination <= 6 ? s.termination : 0]
              << "\n";
    std::cout << "Average point change: " << (work->empty() ? 0.0 : s.mean_point_change * (double)owner.size() / (double)work->size())
              << " (maximum change: " << s.max_point_change << ")" << std::endl;
}

}  // namespace osfm_adapter

#endif /* OSFM_BA_HIP_ADAPTER_HEADER */
