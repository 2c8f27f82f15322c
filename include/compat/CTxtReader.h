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

class CTxtReader {
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
        float *buf = new float[values.size() ? values.size() : 1];
        for (size_t i = 0; i < values.size(); ++i) buf[i] = values[i];
        return flann::Matrix<float>(buf, rows, 3);
    }
};
