// les_types.h -- minimal value types of the host side (the reference uses cv::Rect / cv::Mat / Plane).
// Header-only, no OpenCV.  Names and field meaning follow the reference so that host code written
// against LES/*.h reads the same: Rect{x,y,width,height}, Plane{a,b,c,v}, Parameters (LES/StereoEnergy.h:13-40).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace les_host {

struct Point { int x = 0, y = 0; };

struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() = default;
    Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
    Point tl() const { return Point{x, y}; }
    int area() const { return width * height; }
    bool empty() const { return width <= 0 || height <= 0; }
};
// intersection (cv::Rect operator&)
inline Rect operator&(const Rect& a, const Rect& b)
{
    const int x1 = std::max(a.x, b.x), y1 = std::max(a.y, b.y);
    const int x2 = std::min(a.x + a.width, b.x + b.width), y2 = std::min(a.y + a.height, b.y + b.height);
    return (x2 > x1 && y2 > y1) ? Rect(x1, y1, x2 - x1, y2 - y1) : Rect();
}

// struct Plane of LES/Plane.h: disparity z = a*x + b*y + c; v = vertical disparity (always 0 here)
struct Plane {
    float a = 0, b = 0, c = 0, v = 0;
    Plane() = default;
    Plane(float a_, float b_, float c_, float v_ = 0) : a(a_), b(b_), c(c_), v(v_) {}
    // plane with unit normal (nx,ny,nz) through disparity z at pixel (x,y)          (LES/Plane.h:14-40)
    static Plane CreatePlane(float nx, float ny, float nz, float z, float x, float y, float v = 0)
    {
        Plane p;
        p.a = -nx / nz;
        p.b = -ny / nz;
        p.c = z - p.a * x - p.b * y;
        p.v = v;
        return p;
    }
    void GetNormal(float n[3]) const                                                 // LES/Plane.h:42-50
    {
        const float nz = float(1.0 / std::sqrt(1.0 + a * a + b * b));
        n[0] = -a * nz; n[1] = -b * nz; n[2] = nz;
    }
    float GetZ(float x, float y) const { return a * x + b * y + c; }                 // LES/Plane.h:51-54
    bool operator==(const Plane& o) const { return a == o.a && b == o.b && c == o.c && v == o.v; }
};
static_assert(sizeof(Plane) == 16, "Plane must stay a 16-byte POD (label maps are H x W x 4 float)");

// LES/StereoEnergy.h:13-40 (only the members the matching-cost path reads are kept meaningful)
struct Parameters {
    float alpha = 0.9f, omega = 10.0f, th_grad = 2.0f, th_col = 10.0f, lambda = 20, th_smooth = 1.0f, epsilon = 0.01f;
    float filter_param1 = 10;
    int windR = 20, neighborNum = 8;
    std::string filterName = "GF";
    Parameters(float lambda_ = 20, int windR_ = 20, std::string filterName_ = "GF", float filter_param1_ = 10)
        : lambda(lambda_), filter_param1(filter_param1_), windR(windR_), filterName(std::move(filterName_)) {}
};

// A caller-owned row-major float map (the cv::Mat `costs` / `currentCost_` of the reference).
struct CostMap {
    std::vector<float> data;
    int rows = 0, cols = 0;
    CostMap() = default;
    CostMap(int r, int c, float fill = 0.f) : data((size_t)r * c, fill), rows(r), cols(c) {}
    float& at(int y, int x) { return data[(size_t)y * cols + x]; }
    float at(int y, int x) const { return data[(size_t)y * cols + x]; }
    // pointer to element (r.y, r.x): the view map(r) whose row stride stays `cols`
    float* view(const Rect& r) { return data.data() + (size_t)r.y * cols + r.x; }
};
struct LabelMap {
    std::vector<Plane> data;
    int rows = 0, cols = 0;
    LabelMap() = default;
    LabelMap(int r, int c) : data((size_t)r * c), rows(r), cols(c) {}
    Plane& at(int y, int x) { return data[(size_t)y * cols + x]; }
    const Plane& at(int y, int x) const { return data[(size_t)y * cols + x]; }
};

// cv::RNG [recollection of OpenCV 3.1]: multiply-with-carry generator used by the reference through
// cv::theRNG() (LES/Proposer.h:39,132; LES/Utilities.hpp:256-257; LES/StereoEnergy.h:122).
struct RNG {
    uint64_t state;
    explicit RNG(uint64_t s = 0xffffffffULL) : state(s ? s : 0xffffffffULL) {}
    uint32_t next()
    {
        state = (uint64_t)(uint32_t)state * 4164903690ULL + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a) + a); }
    float uniform(float a, float b) { return (next() * 2.3283064365386962890625e-10f) * (b - a) + a; }
    double uniform(double a, double b)
    {
        const uint32_t t = next();
        const double d = (double)(((uint64_t)t << 32) | next()) * 5.4210108624275221700372640043497e-20;
        return d * (b - a) + a;
    }
};

}  // namespace les_host
