// The reference's driver (PointCloud/Registration/test.cpp:14-29) against the
// drop-in classes: only the include lines differ.
// Build: make -C stereovision_amd/csrc example     Run: ./examples/test_icp 17.txt 15.txt
// A third argument "p2p" runs the same sequence with CPoint2PointICP (the class test.cpp
// includes next to the one it uses).
#include <chrono>
#include <iostream>
#include <string>

#include "CPoint2PlaneICP.h"
#include "CPoint2PointICP.h"
#include "CTxtReader.h"

template <class Icp>
int drive(flann::Matrix<float> dataset, flann::Matrix<float> query);

int main(int argc, char **argv)
{
    CTxtReader reader;
    flann::Matrix<float> dataset = reader.read(argc > 1 ? argv[1] : "17.txt");
    flann::Matrix<float> query = reader.read(argc > 2 ? argv[2] : "15.txt");
    if (argc > 3 && std::string(argv[3]) == "p2p") return drive<CPoint2PointICP>(dataset, query);
    return drive<CPoint2PlaneICP>(dataset, query);
}

template <class Icp>
int drive(flann::Matrix<float> dataset, flann::Matrix<float> query)
{
    Icp icp;
    icp.setReferenceData(dataset);
    auto t1 = std::chrono::steady_clock::now();
    icp.setMoveData(query);
    icp.setMaxIterateCount(5);
    icp.run();
    auto t2 = std::chrono::steady_clock::now();
    double dr_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    std::cout << "searchint spend " << dr_ms << " ms" << std::endl;
    if (icp.lastStatus() != 0) {
        std::cerr << "svicp: " << icp.lastError() << std::endl;
        return 1;
    }
    return 0;
}
