// Minimal stand-in for flann::Matrix<T>, the view type the reference's
// registration classes take (used at PointCloud/Registration/CPoint2PlaneICP.h:13-14,
// CKNNSearch.h:14-17, built at CTxtReader.cpp:57).  FLANN itself is not needed:
// only the non-owning row-major view crosses the class boundary.
#pragma once
#include <cstddef>

namespace flann {

template <typename T>
class Matrix {
public:
    size_t rows, cols, stride;
    Matrix() : rows(0), cols(0), stride(0), data_(nullptr) {}
    Matrix(T *data, size_t rows_, size_t cols_, size_t stride_ = 0)
        : rows(rows_), cols(cols_), stride(stride_ ? stride_ : sizeof(T) * cols_), data_(data) {}
    T *operator[](size_t row) const
    {
        return reinterpret_cast<T *>(reinterpret_cast<unsigned char *>(data_) + row * stride);
    }
    T *ptr() const { return data_; }

private:
    T *data_;
};

}  // namespace flann
