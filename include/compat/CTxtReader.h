// Drop-in for the reference's CTxtReader (PointCloud/Registration/CTxtReader.h,
// CTxtReader.cpp:13-61): space-separated numbers, atof -> float, three per row.
// The returned view owns nothing; like the reference, the buffer is the caller's
// to keep (the reference never frees it).
#pragma once
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "flann_matrix.hpp"
#include "svicp.h"

class CTxtReader {
    // The clouds are handed to the GPU (and the move data back) on every run(): they are
    // allocated in pinned host memory when the library can provide it, so that those copies
    // run at the speed of the link; plain new[] otherwise.  Never freed, as in the reference
    // (CTxtReader.cpp:52).
    static float *cloud_buffer(size_t floats)
    {
        void *p = nullptr;
        if (floats >= (1u << 18) && svicp_host_alloc((int64_t)(floats * sizeof(float)), &p) == SVICP_OK && p)
            return static_cast<float *>(p);
        return new float[floats ? floats : 1];
    }

public:
    CTxtReader() {}
    virtual ~CTxtReader() {}
    flann::Matrix<float> read(std::string path)
    {
        std::ifstream f(path);
        flann::Matrix<float> data;
        if (!f.is_open()) return data;
        std::vector<float> values;
        std::string line;
        while (std::getline(f, line)) {
            size_t off = 0;
            while (off <= line.size()) {
                size_t sp = line.find(' ', off);
                if (sp == std::string::npos) sp = line.size();
                if (sp > off) values.push_back((float)std::atof(line.substr(off, sp - off).c_str()));
                off = sp + 1;
            }
        }
        size_t rows = values.size() / 3;
        float *buf = cloud_buffer(values.size());
        for (size_t i = 0; i < values.size(); ++i) buf[i] = values[i];
        return flann::Matrix<float>(buf, rows, 3);
    }

    // Addition (no counterpart in the reference): the same cloud as raw little-endian
    // float32 x y z triples, for clouds of many millions of points where parsing text takes
    // longer than registering.  Same ownership rule as read().
    flann::Matrix<float> readBinary(std::string path)
    {
        std::ifstream f(path, std::ios::binary | std::ios::ate);
        flann::Matrix<float> data;
        if (!f.is_open()) return data;
        const std::streamoff bytes = f.tellg();
        const size_t rows = (size_t)bytes / (3 * sizeof(float));
        float *buf = cloud_buffer(rows * 3);
        f.seekg(0);
        f.read(reinterpret_cast<char *>(buf), (std::streamsize)(rows * 3 * sizeof(float)));
        return flann::Matrix<float>(buf, rows, 3);
    }

    static bool writeBinary(std::string path, flann::Matrix<float> cloud)
    {
        std::ofstream f(path, std::ios::binary);
        if (!f.is_open()) return false;
        f.write(reinterpret_cast<const char *>(cloud.ptr()), (std::streamsize)(cloud.rows * 3 * sizeof(float)));
        return (bool)f;
    }
};
