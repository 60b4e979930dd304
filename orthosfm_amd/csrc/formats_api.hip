// On-disk text formats of the reference pipeline behind the C ABI.
// Host code only.  Every number goes through the C++ stream operation the reference
// applies to it (ostream << float / double / unsigned, std::to_string(double),
// std::stoi / stof / stod on the way back), so the bytes are the reference's by
// construction when both sides are built against the same standard library.
//
//   tracks.txt             src/matching/matching_io.cpp:16-48 (write), :50-97 (read)
//   %03d_%03d.txt          src/matching/matching_io.cpp:99-141
//   cameras.txt            src/data_structures/camera_io.cpp:15-40 (write), :42-71 (read)
//   sparse_cloud.ply       src/util/common.cpp:141-188
//   time_measurements.txt  src/util/timing.cpp:18-28 (write), :30-53 (read)
//   MVE -> orthosfm tracks src/matching/matching_mve.cpp:455-466
#include <algorithm>
#include <cstring>
#include <fstream>
#include <iterator>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "osfm_common.h"

using namespace osfm;

namespace {

// boost::split(out, line, boost::is_any_of(sep)) without token compression
void split(const std::string &line, char sep, std::vector<std::string> &out)
{
    out.clear();
    size_t from = 0;
    for (;;) {
        const size_t at = line.find(sep, from);
        if (at == std::string::npos) { out.push_back(line.substr(from)); return; }
        out.push_back(line.substr(from, at - from));
        from = at + 1;
    }
}

bool csr_ok(const char *who, int64_t num_tracks, const int64_t *offs, const void *features)
{
    if (num_tracks < 0 || (num_tracks > 0 && !offs)) {
        set_error("%s: null offsets / negative track count", who);
        return false;
    }
    for (int64_t t = 0; t < num_tracks; ++t)
        if (offs[t] < 0 || offs[t + 1] < offs[t]) {
            set_error("%s: track %lld has the feature range [%lld, %lld)", who, (long long)t,
                (long long)offs[t], (long long)offs[t + 1]);
            return false;
        }
    if (num_tracks > 0 && offs[num_tracks] > 0 && !features) {
        set_error("%s: features is null", who);
        return false;
    }
    return true;
}

// zfill (src/util/common.cpp:40-48)
std::string zfill(int value, size_t zeros)
{
    std::string str = std::to_string(value);
    if (zeros > str.size()) str.insert(0, zeros - str.size(), '0');
    return str;
}

}  // namespace

extern "C" {

OSFM_API int osfm_tracks_file_write(const char *path, int64_t num_tracks,
    const int64_t *track_offsets, const osfm_track_feature *features)
{
    if (!path) { set_error("tracks_file_write: path is null"); return OSFM_E_ARG; }
    if (!csr_ok("tracks_file_write", num_tracks, track_offsets, features)) return OSFM_E_ARG;
    std::ofstream outfile;
    outfile.open(path);
    if (!outfile) { set_error("tracks_file_write: cannot open %s", path); return OSFM_E_IO; }
    for (int64_t t = 0; t < num_tracks; ++t) {
        const int64_t n = track_offsets[t + 1] - track_offsets[t];
        outfile << (size_t)n << ";";                                         // :25
        for (int64_t k = 0; k < n; ++k) {
            const osfm_track_feature &f = features[track_offsets[t] + k];
            outfile << f.view_id << ";";
            outfile << f.local_feature_id << ";";
            outfile << f.global_feature_id << ";";
            outfile << f.x << ";";
            outfile << f.y << ";";
            outfile << f.r << ";";
            outfile << f.g << ";";
            outfile << f.b;
            if (k < n - 1) outfile << ";";
        }
        outfile << "\n";
    }
    outfile.close();
    if (!outfile) { set_error("tracks_file_write: write to %s failed", path); return OSFM_E_IO; }
    return OSFM_OK;
}

OSFM_API int osfm_tracks_file_read(const char *path, int64_t track_capacity,
    int64_t feature_capacity, int64_t *track_offsets, osfm_track_feature *features,
    int64_t *num_tracks, int64_t *num_features)
{
    if (!path || !num_tracks || !num_features || track_capacity < 0 || feature_capacity < 0) {
        set_error("tracks_file_read: null argument / negative capacity");
        return OSFM_E_ARG;
    }
    std::ifstream file(path);
    if (!file) { set_error("tracks_file_read: cannot open %s", path); return OSFM_E_IO; }
    std::vector<int64_t> offs(1, 0);
    std::vector<osfm_track_feature> feats;
    std::string line;
    std::vector<std::string> s;
    int64_t lineno = 0;
    while (std::getline(file, line)) {
        ++lineno;
        split(line, ';', s);
        try {
            const int count = std::stoi(s[0]);                               // :67
            if (count < 0 || (size_t)count * 8 + 1 > s.size()) throw std::out_of_range("fields");
            size_t at = 1;
            for (int k = 0; k < count; ++k, at += 8) {
                osfm_track_feature f;
                f.view_id = (uint32_t)std::stoi(s[at]);
                f.local_feature_id = (uint32_t)std::stoi(s[at + 1]);
                f.global_feature_id = (uint32_t)std::stoi(s[at + 2]);
                f.x = std::stof(s[at + 3]);
                f.y = std::stof(s[at + 4]);
                f.r = (uint32_t)std::stoi(s[at + 5]);
                f.g = (uint32_t)std::stoi(s[at + 6]);
                f.b = (uint32_t)std::stoi(s[at + 7]);
                feats.push_back(f);
            }
        } catch (const std::exception &e) {
            set_error("tracks_file_read: %s line %lld does not parse (%s)", path, (long long)lineno, e.what());
            return OSFM_E_IO;
        }
        offs.push_back((int64_t)feats.size());
    }
    *num_tracks = (int64_t)offs.size() - 1;
    *num_features = (int64_t)feats.size();
    if (*num_tracks > track_capacity || *num_features > feature_capacity) {
        set_error("tracks_file_read: %lld tracks / %lld features, capacity %lld / %lld",
            (long long)*num_tracks, (long long)*num_features, (long long)track_capacity,
            (long long)feature_capacity);
        return OSFM_E_CAPACITY;
    }
    if (!track_offsets || (!feats.empty() && !features)) {
        set_error("tracks_file_read: output array is null");
        return OSFM_E_ARG;
    }
    std::memcpy(track_offsets, offs.data(), offs.size() * sizeof(int64_t));
    if (!feats.empty()) std::memcpy(features, feats.data(), feats.size() * sizeof(osfm_track_feature));
    return OSFM_OK;
}

OSFM_API int osfm_tracks_pairwise_files_write(const char *folder, int32_t num_views,
    const uint32_t *view_ids, int64_t num_tracks, const int64_t *track_offsets,
    const osfm_track_feature *features, int64_t *files_written)
{
    if (!folder || num_views < 0 || (num_views > 0 && !view_ids)) {
        set_error("tracks_pairwise_files_write: null argument / negative view count");
        return OSFM_E_ARG;
    }
    if (!csr_ok("tracks_pairwise_files_write", num_tracks, track_offsets, features)) return OSFM_E_ARG;
    if (files_written) *files_written = 0;
    // The reference filters the whole track list once per pair (V^2 T feature visits).
    // Only a track with a feature of view i or view j can hold two features of {i, j},
    // so a pair walks the merged (track-ordered) lists of its two views: same files.
    std::unordered_map<uint32_t, std::vector<int64_t>> tracks_of_view;
    for (int64_t t = 0; t < num_tracks; ++t)
        for (int64_t k = track_offsets[t]; k < track_offsets[t + 1]; ++k) {
            std::vector<int64_t> &l = tracks_of_view[features[k].view_id];
            if (l.empty() || l.back() != t) l.push_back(t);
        }
    const std::vector<int64_t> none;
    std::vector<int64_t> hold;
    for (int32_t i = 0; i < num_views; ++i) {
        const auto li = tracks_of_view.find(view_ids[i]);
        const std::vector<int64_t> &ti = li == tracks_of_view.end() ? none : li->second;
        for (int32_t j = i + 1; j < num_views; ++j) {
            const auto lj = tracks_of_view.find(view_ids[j]);
            const std::vector<int64_t> &tj = lj == tracks_of_view.end() ? none : lj->second;
            hold.clear();
            std::set_union(ti.begin(), ti.end(), tj.begin(), tj.end(), std::back_inserter(hold));
            const uint32_t ids[2] = {view_ids[i], view_ids[j]};
            std::ofstream outfile;
            bool open = false;
            for (const int64_t t : hold) {
                // filterTracksToAvailableCameras(ids, tracks, true, false): the features of
                // the two views, kept when there are exactly ids.size() == 2 of them
                int64_t cur[3];
                int n = 0;
                for (int64_t k = track_offsets[t]; k < track_offsets[t + 1] && n < 3; ++k)
                    if (features[k].view_id == ids[0] || features[k].view_id == ids[1]) cur[n++] = k;
                if (n != 2) continue;
                if (!open) {
                    const std::string file_path =
                        std::string(folder) + "/" + zfill((int)ids[0], 3) + "_" + zfill((int)ids[1], 3) + ".txt";
                    outfile.open(file_path);
                    if (!outfile) {
                        set_error("tracks_pairwise_files_write: cannot open %s", file_path.c_str());
                        return OSFM_E_IO;
                    }
                    open = true;
                    if (files_written) ++*files_written;
                }
                for (int k = 0; k < 2; ++k)                                  // :124-139
                    for (int c = 0; c < 2; ++c) {
                        const osfm_track_feature &f = features[cur[c]];
                        if (f.view_id == ids[k]) {
                            outfile << f.x << " " << f.y;
                            if (k == 0) outfile << " "; else outfile << "\n";
                        }
                    }
            }
            if (open) {
                outfile.close();
                if (!outfile) { set_error("tracks_pairwise_files_write: write failed"); return OSFM_E_IO; }
            }
        }
    }
    return OSFM_OK;
}

OSFM_API int osfm_tracks_from_mve(int64_t num_features, const int32_t *track_features,
    int32_t num_views, const int64_t *view_starts, const float *positions,
    const uint8_t *colors, double image_width, osfm_track_feature *features)
{
    if (num_features < 0 || num_views < 0 || !view_starts ||
        (num_features > 0 && (!track_features || !positions || !features))) {
        set_error("tracks_from_mve: null argument / negative count");
        return OSFM_E_ARG;
    }
    for (int64_t k = 0; k < num_features; ++k) {
        const int32_t v = track_features[2 * k], f = track_features[2 * k + 1];
        if (v < 0 || v >= num_views || f < 0 || f >= view_starts[v + 1] - view_starts[v]) {
            set_error("tracks_from_mve: feature %lld names (%d, %d)", (long long)k, v, f);
            return OSFM_E_RANGE;
        }
        const int64_t row = view_starts[v] + f;
        osfm_track_feature &o = features[k];
        o.view_id = (uint32_t)v;
        o.local_feature_id = (uint32_t)f;
        o.global_feature_id = (uint32_t)(32768 * v + f);
        // imageWidth * (pos[0] + 0.5): float + double -> double, stored in a float member
        o.x = (float)(image_width * (positions[2 * row] + 0.5));
        o.y = (float)(image_width * (positions[2 * row + 1] + 0.5));
        o.r = colors ? colors[3 * row] : 0;
        o.g = colors ? colors[3 * row + 1] : 0;
        o.b = colors ? colors[3 * row + 2] : 0;
    }
    return OSFM_OK;
}

OSFM_API int osfm_cameras_file_write(const char *path, int32_t num_cameras,
    const char *const *image_names, const double *matrices)
{
    if (!path || num_cameras < 0 || (num_cameras > 0 && (!image_names || !matrices))) {
        set_error("cameras_file_write: null argument / negative count");
        return OSFM_E_ARG;
    }
    std::ofstream outfile;
    outfile.open(path);
    if (!outfile) { set_error("cameras_file_write: cannot open %s", path); return OSFM_E_IO; }
    for (int32_t c = 0; c < num_cameras; ++c) {
        if (!image_names[c]) { set_error("cameras_file_write: name %d is null", c); return OSFM_E_ARG; }
        const double *m = matrices + 16 * (size_t)c;
        outfile << std::string(image_names[c]) + ";";
        for (int r = 0; r < 4; ++r) {
            std::string row;
            for (int k = 0; k < 4; ++k) {
                row += std::to_string(m[4 * r + k]);
                if (r < 3 || k < 3) row += ",";
            }
            outfile << row;
        }
        outfile << "\n";
    }
    outfile.close();
    if (!outfile) { set_error("cameras_file_write: write to %s failed", path); return OSFM_E_IO; }
    return OSFM_OK;
}

OSFM_API int osfm_cameras_file_read(const char *path, int32_t camera_capacity,
    int64_t names_capacity, char *names_buf, double *matrices, int32_t *num_cameras,
    int64_t *names_bytes)
{
    if (!path || !num_cameras || !names_bytes || camera_capacity < 0 || names_capacity < 0) {
        set_error("cameras_file_read: null argument / negative capacity");
        return OSFM_E_ARG;
    }
    std::ifstream file(path);
    if (!file) { set_error("cameras_file_read: cannot open %s", path); return OSFM_E_IO; }
    std::string names;
    std::vector<double> mats;
    std::string line;
    std::vector<std::string> parts, s;
    int32_t n = 0;
    while (std::getline(file, line)) {
        split(line, ';', parts);
        try {
            if (parts.size() < 2) throw std::out_of_range("no ';'");
            split(parts[1], ',', s);
            if (s.size() < 16) throw std::out_of_range("fewer than 16 entries");
            for (int k = 0; k < 16; ++k) mats.push_back(std::stod(s[k]));
        } catch (const std::exception &e) {
            set_error("cameras_file_read: %s line %d does not parse (%s)", path, n + 1, e.what());
            return OSFM_E_IO;
        }
        names += parts[0];
        names.push_back('\0');
        ++n;
    }
    *num_cameras = n;
    *names_bytes = (int64_t)names.size();
    if (n > camera_capacity || (int64_t)names.size() > names_capacity) {
        set_error("cameras_file_read: %d cameras / %lld name bytes, capacity %d / %lld", n,
            (long long)names.size(), camera_capacity, (long long)names_capacity);
        return OSFM_E_CAPACITY;
    }
    if (n > 0 && (!names_buf || !matrices)) { set_error("cameras_file_read: output array is null"); return OSFM_E_ARG; }
    if (n > 0) {
        std::memcpy(names_buf, names.data(), names.size());
        std::memcpy(matrices, mats.data(), mats.size() * sizeof(double));
    }
    return OSFM_OK;
}

OSFM_API int osfm_sparse_cloud_write(const char *path, int64_t num_tracks,
    const int64_t *track_offsets, const osfm_track_feature *features, const double *points,
    const uint8_t *has_point)
{
    if (!path || (num_tracks > 0 && (!points || !has_point))) {
        set_error("sparse_cloud_write: null argument");
        return OSFM_E_ARG;
    }
    if (!csr_ok("sparse_cloud_write", num_tracks, track_offsets, features)) return OSFM_E_ARG;
    int points_count = 0;
    for (int64_t t = 0; t < num_tracks; ++t)
        if (has_point[t]) {
            if (track_offsets[t + 1] == track_offsets[t]) {
                set_error("sparse_cloud_write: track %lld has a point but no feature", (long long)t);
                return OSFM_E_ARG;
            }
            ++points_count;
        }
    std::ofstream filestream;
    filestream.open(path, std::ios::out | std::ios::trunc);               // the reference removes the old file first
    if (!filestream) { set_error("sparse_cloud_write: cannot open %s", path); return OSFM_E_IO; }
    filestream << "ply" << "\n";
    filestream << "format ascii 1.0" << "\n";
    filestream << "element vertex " << points_count << "\n";
    filestream << "property float x" << "\n";
    filestream << "property float y" << "\n";
    filestream << "property float z" << "\n";
    filestream << "property uchar red" << "\n";
    filestream << "property uchar green" << "\n";
    filestream << "property uchar blue" << "\n";
    filestream << "end_header" << "\n";
    for (int64_t t = 0; t < num_tracks; ++t) {
        if (!has_point[t]) continue;
        const double *p = points + 4 * (size_t)t;
        const osfm_track_feature &f0 = features[track_offsets[t]];
        filestream << p[0] << " ";
        filestream << p[1] << " ";
        filestream << p[2] << " ";
        filestream << (int)f0.r << " ";
        filestream << (int)f0.g << " ";
        filestream << (int)f0.b << "\n";
    }
    filestream.close();
    if (!filestream) { set_error("sparse_cloud_write: write to %s failed", path); return OSFM_E_IO; }
    return OSFM_OK;
}

OSFM_API int osfm_time_measurements_write(const char *path, const double *seconds)
{
    if (!path || !seconds) { set_error("time_measurements_write: null argument"); return OSFM_E_ARG; }
    std::ofstream outfile;
    outfile.open(path);
    if (!outfile) { set_error("time_measurements_write: cannot open %s", path); return OSFM_E_IO; }
    outfile << "Initialization Time [s] = " << seconds[0] << "\n";
    outfile << "Track Building Time [s] = " << seconds[1] << "\n";
    outfile << "Pose Estimation Time [s] = " << seconds[2] << "\n";
    outfile << "Total Time [s] = " << seconds[3] << "\n";
    outfile.close();
    if (!outfile) { set_error("time_measurements_write: write to %s failed", path); return OSFM_E_IO; }
    return OSFM_OK;
}

OSFM_API int osfm_time_measurements_read(const char *path, double *seconds)
{
    if (!path || !seconds) { set_error("time_measurements_read: null argument"); return OSFM_E_ARG; }
    std::ifstream file(path);
    if (!file) { set_error("time_measurements_read: cannot open %s", path); return OSFM_E_IO; }
    std::string line;
    std::vector<std::string> s;
    int current = 0;
    for (int k = 0; k < 4; ++k) seconds[k] = 0.0;
    while (std::getline(file, line)) {
        split(line, '=', s);
        try {
            if (s.size() < 2) throw std::out_of_range("no '='");
            if (current < 4) seconds[current] = std::stod(s[1]);
        } catch (const std::exception &e) {
            set_error("time_measurements_read: %s line %d does not parse (%s)", path, current + 1, e.what());
            return OSFM_E_IO;
        }
        ++current;
    }
    return OSFM_OK;
}

}  // extern "C"
