"""Generator of tests/golden/cfg1_suzanne.npz -- BASELINE configs[0], "3-image synthetic subset
(Suzanne)".  Runs in the BUILD container only (it reads /root/reference, which does not exist on
the GPU box); the .npz it writes is data: the landmark positions of the reference's own test model
and the camera angles its test bench would draw.

* points: the vertices of /root/reference/resources/Suzanne.ply, parsed the way
  orthosfm::loadPointClouds does (src/testbench/dataset_generation.cpp:96-137): every line behind
  "end_header", the first three blank-separated tokens through std::stod -> float64.  All 7872
  rows are kept, duplicates included (the exporter repeats a vertex per face): the test bench
  makes one track per row.
* cams_deg: (phi, theta, roll) in degrees of the 16 cameras of generateGroundTruthCameras
  (dataset_generation.cpp:14-38): camera 0 is (0, 0, 0), camera i sits at phi = 22.5 i with
  theta / roll from std::uniform_real_distribution<double>(-30, 30) on a default-seeded
  std::default_random_engine.  The two draws are arguments of one call, so their order is the
  compiler's; a few lines of C++ with the same call shape (std::make_shared of a type with that
  constructor) are compiled here with g++ and libstdc++ -- the toolchain the reference is built
  with -- and print the values.  None of the reference's code is compiled or copied.
BASELINE configs[0] uses the first three cameras.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

CPP = r"""
#include <cstdio>
#include <memory>
#include <random>
struct Cam { double phi, theta, roll; Cam(int, double p, double t, double r) : phi(p), theta(t), roll(r) {} };
int main() {
    std::uniform_real_distribution<double> thetaRnd(-30, 30);
    std::uniform_real_distribution<double> rollRnd(-30, 30);
    std::default_random_engine re;
    for (int i = 0; i < 16; i++) {
        std::shared_ptr<Cam> cam = (i == 0) ? std::make_shared<Cam>(i, 0, 0, 0)
                                            : std::make_shared<Cam>(i, 22.5 * i, thetaRnd(re), rollRnd(re));
        std::printf("%.17g %.17g %.17g\n", cam->phi, cam->theta, cam->roll);
    }
    return 0;
}
"""


def main():
    pts = []
    header = True
    with open(os.path.join(REF, "resources", "Suzanne.ply")) as f:
        for line in f:
            if header:
                if line.startswith("end_header"):
                    header = False
                continue
            tok = line.rstrip("\n").split(" ")
            pts.append([float(tok[0]), float(tok[1]), float(tok[2])])
    pts = np.array(pts, dtype=np.float64)
    assert pts.shape == (7872, 3), pts.shape
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "cams.cc"), os.path.join(d, "cams")
        open(src, "w").write(CPP)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src])
        out = subprocess.check_output([exe], text=True)
    cams = np.array([[float(x) for x in l.split()] for l in out.strip().splitlines()])
    assert cams.shape == (16, 3) and np.all(np.abs(cams[1:, 1:]) <= 30.0) and np.all(cams[0] == 0)
    np.savez_compressed(os.path.join(HERE, "cfg1_suzanne.npz"), points=pts, cams_deg=cams,
                        width=np.int32(2048), height=np.int32(2048))
    print("cfg1_suzanne.npz:", pts.shape[0], "points, bbox", pts.min(0), pts.max(0))
    print(cams[:4])


if __name__ == "__main__":
    sys.exit(main())
