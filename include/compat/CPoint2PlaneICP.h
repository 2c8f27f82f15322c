// Drop-in for the reference's CPoint2PlaneICP
// (PointCloud/Registration/CPoint2PlaneICP.h:8-27) over the C ABI of libsvicp.so.
// Same public methods and side effects: run() leaves the caller's move buffer
// transformed count-1 times and prints the accumulated transform and its Euler
// angles.  getTransform()/getInfo()/lastStatus() are additions.
#pragma once
#include <cmath>
#include <cstdio>
#include <iostream>

#include "flann_matrix.hpp"
#include "svicp.h"

class CPoint2PlaneICP {
public:
    // the argument is an addition: the GPU the object works on (the reference has no devices)
    explicit CPoint2PlaneICP(int device = 0) : m_eps(0.f), m_count(1), m_ctx(nullptr), m_status(SVICP_OK), m_info()
    {
        m_status = svicp_create(device, &m_ctx);
        for (int i = 0; i < 16; ++i) m_T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    }
    ~CPoint2PlaneICP()
    {
        if (m_pinned) svicp_host_unregister(m_pinned);
        svicp_destroy(m_ctx);
    }
    CPoint2PlaneICP(const CPoint2PlaneICP &) = delete;
    CPoint2PlaneICP &operator=(const CPoint2PlaneICP &) = delete;

    // CPoint2PlaneICP.cpp:11-18: builds the search structure and estimates normals
    void setReferenceData(flann::Matrix<float> ref)
    {
        m_refData = ref;
        m_status = m_ctx ? svicp_set_target(m_ctx, ref.ptr(), (int64_t)ref.rows, nullptr)
                         : SVICP_ERR_NO_DEVICE;
    }
    // addition: caller-supplied unit normals instead of calcNormals()
    void setReferenceData(flann::Matrix<float> ref, flann::Matrix<float> normals)
    {
        m_refData = ref;
        m_status = m_ctx ? svicp_set_target(m_ctx, ref.ptr(), (int64_t)ref.rows, normals.ptr())
                         : SVICP_ERR_NO_DEVICE;
    }
    // :19-22, a view.  run() copies the buffer to the device and back on every call, so a
    // large one is pinned in place here (an addition; it stays the caller's memory).
    // LIFETIME, stricter than the reference's view: a pinned buffer must stay allocated until
    // this object is destroyed or setMoveData is given another buffer (the registration is
    // undone there, and undoing it on freed memory is undefined).  A caller that frees or
    // reallocates its cloud earlier calls setPinMoveData(false) before setMoveData: the copies
    // then take the pageable path and the object keeps nothing but the view.
    void setPinMoveData(bool pin) { m_pin = pin; }
    void setMoveData(flann::Matrix<float> move)
    {
        if (m_pinned) { svicp_host_unregister(m_pinned); m_pinned = nullptr; }
        m_moveData = move;
        const int64_t bytes = (int64_t)move.rows * 3 * (int64_t)sizeof(float);
        if (m_pin && m_ctx && move.ptr() && bytes >= (1 << 20) &&
            svicp_host_register(move.ptr(), bytes) == SVICP_OK)
            m_pinned = move.ptr();
    }
    void setEpsilon(float eps) { m_eps = eps; }                          // :23-26, never read
    void setMaxIterateCount(int count) { m_count = count; }              // :27-30

    // CPoint2PlaneICP.cpp:31-62
    void run()
    {
        if (!m_ctx) { m_status = SVICP_ERR_NO_DEVICE; return; }
        m_status = svicp_set_source(m_ctx, m_moveData.ptr(), (int64_t)m_moveData.rows);
        if (m_status) return;
        m_status = svicp_run(m_ctx, m_count, m_T, &m_info);
        // the reference works in the caller's buffer; hand the updated cloud back
        if (m_moveData.rows) svicp_get_source(m_ctx, m_moveData.ptr());
        std::cout << "Current Transformation:" << std::endl;
        for (int r = 0; r < 4; ++r)
            std::cout << m_T[4 * r] << " " << m_T[4 * r + 1] << " " << m_T[4 * r + 2] << " "
                      << m_T[4 * r + 3] << std::endl;
        double ea[3];
        eulerXYZ(ea);
        std::cout << "Euler Angle X Y Z: \n"
                  << ea[0] / 3.1415926 * 180.0 << "\n" << ea[1] / 3.1415926 * 180.0 << "\n"
                  << ea[2] / 3.1415926 * 180.0 << std::endl;
    }

    const double *getTransform() const { return m_T; }        // row-major 4x4
    const svicp_info &getInfo() const { return m_info; }
    int lastStatus() const { return m_status; }
    const char *lastError() const { return svicp_last_error(m_ctx); }
    svicp_ctx *context() { return m_ctx; }

private:
    // Eigen's Matrix3d::eulerAngles(0, 1, 2) (CPoint2PlaneICP.cpp:60): angles a, b, c
    // with R = Rx(a) * Ry(b) * Rz(c), a in [0, pi]
    void eulerXYZ(double out[3]) const
    {
        auto R = [&](int r, int c) { return m_T[4 * r + c]; };
        const double pi = 3.14159265358979323846;
        double a = std::atan2(R(1, 2), R(2, 2));
        double c2 = std::hypot(R(0, 0), R(0, 1));
        double b;
        if (a > 0) {
            a -= pi;
            b = std::atan2(-R(0, 2), -c2);
        } else {
            b = std::atan2(-R(0, 2), c2);
        }
        double s1 = std::sin(a), c1 = std::cos(a);
        double c = std::atan2(s1 * R(2, 0) - c1 * R(1, 0), c1 * R(1, 1) - s1 * R(2, 1));
        out[0] = -a; out[1] = -b; out[2] = -c;
    }

    flann::Matrix<float> m_refData, m_moveData;
    float m_eps;
    int m_count;
    svicp_ctx *m_ctx;
    float *m_pinned = nullptr;
    bool m_pin = true;
    int m_status;
    svicp_info m_info;
    double m_T[16];
};
