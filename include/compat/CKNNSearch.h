// Drop-in for the reference's CKNNSearch (PointCloud/Registration/CKNNSearch.h:9-25)
// over the C ABI of libsvicp.so.  Same public methods; the FLANN kd-tree is
// replaced by the device grid inside the library.
#pragma once
#include <cmath>
#include <utility>

#include "flann_matrix.hpp"
#include "svicp.h"

class CKNNSearch {
public:
    // the argument is an addition: the GPU the object works on (the reference has no devices)
    explicit CKNNSearch(int device = 0) : m_k(1), m_ctx(nullptr), m_curRows(0), m_status(SVICP_OK)
    {
        m_status = svicp_create(device, &m_ctx);
    }
    virtual ~CKNNSearch()
    {
        svicp_destroy(m_ctx);
        delete[] m_indices.ptr();
        delete[] m_dists.ptr();
    }
    CKNNSearch(const CKNNSearch &) = delete;
    CKNNSearch &operator=(const CKNNSearch &) = delete;

    // CKNNSearch.cpp:27-35 (index build)
    void setTreeData(flann::Matrix<float> &data)
    {
        m_treeData = data;
        // no normals are needed for a pure search: pass the points as a dummy array
        m_status = m_ctx ? svicp_set_target(m_ctx, data.ptr(), (int64_t)data.rows, data.ptr())
                         : SVICP_ERR_NO_DEVICE;
    }
    // CKNNSearch.cpp:37-54 (result buffers grow, never shrink)
    void setQueryData(flann::Matrix<float> &querydata)
    {
        m_queryData = querydata;
        if ((int)m_queryData.rows > m_curRows) {
            delete[] m_indices.ptr();
            delete[] m_dists.ptr();
            m_indices = flann::Matrix<int>(new int[m_queryData.rows * m_k], m_queryData.rows, m_k);
            m_dists = flann::Matrix<float>(new float[m_queryData.rows * m_k], m_queryData.rows, m_k);
            m_curRows = (int)m_queryData.rows;
        }
    }
    // CKNNSearch.cpp:60-65
    void setKCount(int count) { m_k = count > 128 ? 128 : count; }

    // CKNNSearch.cpp:67-71: views into buffers owned by this object
    std::pair<flann::Matrix<int>, flann::Matrix<float>> getKNeighbour()
    {
        if (m_ctx && m_indices.ptr() && m_indices.cols == (size_t)m_k)
            m_status = svicp_knn(m_ctx, m_queryData.ptr(), (int64_t)m_queryData.rows, m_k,
                                 m_indices.ptr(), m_dists.ptr());
        else
            m_status = SVICP_ERR_ARG;   // setKCount after setQueryData: buffers have the old width
        return std::make_pair(m_indices, m_dists);
    }

    // additions: the reference reports nothing
    int lastStatus() const { return m_status; }
    const char *lastError() const { return svicp_last_error(m_ctx); }

private:
    int m_k;
    flann::Matrix<float> m_treeData, m_queryData;
    flann::Matrix<int> m_indices;
    flann::Matrix<float> m_dists;
    svicp_ctx *m_ctx;
    int m_curRows;
    int m_status;
};
