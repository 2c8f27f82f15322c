// Registers one cloud onto another through the drop-in classes, the way a program written
// against the reference's PointCloud/Registration classes does (the call sequence of
// PointCloud/Registration/test.cpp:16-24: read two clouds, setReferenceData, setMoveData,
// setMaxIterateCount, run).
//
//   make -C stereovision_amd/csrc example
//   ./examples/test_icp <target.txt> <source.txt> [plane|p2p] [iterations]
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <string>

#include "CPoint2PlaneICP.h"
#include "CPoint2PointICP.h"
#include "CTxtReader.h"

namespace {

template <class Registration>
int registerPair(flann::Matrix<float> target, flann::Matrix<float> source, int iterations)
{
    using clock = std::chrono::steady_clock;
    Registration reg;
    reg.setReferenceData(target);                // index (and normals) of the target: not timed
    const clock::time_point begin = clock::now();
    reg.setMoveData(source);
    reg.setMaxIterateCount(iterations);
    reg.run();                                   // prints the transform and its Euler angles
    const double ms = std::chrono::duration<double, std::milli>(clock::now() - begin).count();
    // the reference's driver words its timing line like this (test.cpp:28)
    std::cout << "searchint spend " << ms << " ms" << std::endl;
    if (reg.lastStatus() != 0) {
        std::cerr << "svicp: " << reg.lastError() << std::endl;
        return 1;
    }
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    const std::string targetPath = argc > 1 ? argv[1] : "17.txt";
    const std::string sourcePath = argc > 2 ? argv[2] : "15.txt";
    const std::string mode = argc > 3 ? argv[3] : "plane";
    const int iterations = argc > 4 ? std::atoi(argv[4]) : 5;   // test.cpp:23 runs five
    CTxtReader reader;
    flann::Matrix<float> target = reader.read(targetPath);
    flann::Matrix<float> source = reader.read(sourcePath);
    if (target.rows == 0 || source.rows == 0) {
        std::cerr << "could not read " << (target.rows == 0 ? targetPath : sourcePath) << std::endl;
        return 2;
    }
    return mode == "p2p" ? registerPair<CPoint2PointICP>(target, source, iterations)
                         : registerPair<CPoint2PlaneICP>(target, source, iterations);
}
