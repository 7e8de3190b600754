// cvlite.hpp -- the handful of OpenCV value types that appear in the public surface of
// AnalyzerUnit / L3Localizer / Trainer / bubble / Parser (reference headers include
// <opencv2/opencv.hpp>; OpenCV is not available in this image).  Only types, no image processing:
// every pixel operation of the hot path runs in the HIP library (include/abub_hip.h).
//
// Define ABUB_USE_OPENCV to build the same sources against a real OpenCV instead (the classes only
// rely on the members declared here).
#ifndef ABUB3HS_CVLITE_HPP
#define ABUB3HS_CVLITE_HPP

#ifdef ABUB_USE_OPENCV
#include <opencv2/opencv.hpp>
#else

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

typedef unsigned char uchar;

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0

namespace cv {

template <typename T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<int> Point;
typedef Point_<float> Point2f;

template <typename T>
struct Size_ {
    T width, height;
    Size_() : width(0), height(0) {}
    Size_(T w, T h) : width(w), height(h) {}
};
typedef Size_<int> Size;
typedef Size_<float> Size2f;

template <typename T>
struct Rect_ {
    T x, y, width, height;
    Rect_() : x(0), y(0), width(0), height(0) {}
    Rect_(T x_, T y_, T w, T h) : x(x_), y(y_), width(w), height(h) {}
    T area() const { return width * height; }
};
typedef Rect_<int> Rect;

struct RotatedRect {
    Point2f center;
    Size2f size;
    float angle;
    RotatedRect() : angle(0) {}
};

struct Scalar {
    double val[4];
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
};

// spatial moments up to first order are all the localizer reads (L3Localizer.cpp:405-407, 822-823)
struct Moments {
    double m00, m10, m01, m20, m11, m02, m30, m21, m12, m03;
    Moments() { std::memset(this, 0, sizeof(*this)); }
};

// Reference-counted, continuous, single-channel 8-bit (or float) matrix.
class Mat {
public:
    int rows, cols;
    uchar *data;

    Mat() : rows(0), cols(0), data(nullptr), type_(CV_8U) {}
    Mat(int r, int c, int type) : rows(0), cols(0), data(nullptr), type_(CV_8U) { create(r, c, type); }

    void create(int r, int c, int type)
    {
        if (r == rows && c == cols && type == type_ && buf_ && buf_.use_count() == 1)
            return;
        rows = r;
        cols = c;
        type_ = type;
        buf_.reset(new std::vector<uchar>((size_t)r * c * elemSize()));
        data = buf_->data();
    }
    static Mat zeros(int r, int c, int type)
    {
        Mat m(r, c, type);
        if (m.data)
            std::memset(m.data, 0, (size_t)r * c * m.elemSize());
        return m;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    int channels() const { return 1; }
    size_t elemSize() const { return type_ == CV_32F ? 4 : 1; }
    size_t total() const { return (size_t)rows * cols; }
    bool isContinuous() const { return true; }
    Mat clone() const
    {
        Mat m;
        copyTo(m);
        return m;
    }
    void copyTo(Mat &dst) const
    {
        if (empty()) {
            dst.release();
            return;
        }
        dst.create(rows, cols, type_);
        std::memcpy(dst.data, data, total() * elemSize());
    }
    void release()
    {
        buf_.reset();
        data = nullptr;
        rows = cols = 0;
    }
    template <typename T>
    T &at(int r, int c) { return reinterpret_cast<T *>(data)[(size_t)r * cols + c]; }
    template <typename T>
    const T &at(int r, int c) const { return reinterpret_cast<const T *>(data)[(size_t)r * cols + c]; }
    template <typename T>
    T *ptr(int r = 0) { return reinterpret_cast<T *>(data) + (size_t)r * cols; }
    template <typename T>
    const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data) + (size_t)r * cols; }
    uchar *ptr(int r = 0) { return data + (size_t)r * cols * elemSize(); }
    const uchar *ptr(int r = 0) const { return data + (size_t)r * cols * elemSize(); }

private:
    int type_;
    std::shared_ptr<std::vector<uchar>> buf_;
};

enum { IMREAD_GRAYSCALE = 0 };
// 8-bit grey decode of PNG (grey / palette / RGB(A), 1-16 bit, non-interlaced) and BMP (1/4/8/24/32-bit,
// uncompressed) -- the formats of PICO camera frames (RawParser.cpp:39, ZipParser.cpp:222) and of the
// mask files cam_masks/<series>/camN[_bellows]_mask.bmp (L3Localizer.cpp:980-986).  Empty Mat on failure.
Mat imread(const std::string &path, int flags = IMREAD_GRAYSCALE);
Mat imdecode(const std::vector<uchar> &buf, int flags = IMREAD_GRAYSCALE);
Mat imdecode(const uchar *data, size_t size, int flags = IMREAD_GRAYSCALE);
// (not in OpenCV) decodes an 8-bit grey view of the image straight into `dst` (W * H bytes); false when the data is not a
// W x H image; no per-call heap allocation for PNG (thread-local scratch)
bool imdecodeInto(const uchar *data, size_t size, uchar *dst, int W, int H);
// debug write-out (AnalyzerUnit.cpp:237,354-365; L3Localizer.cpp:236-257,448): 8-bit grey PNG, or 8-bit palettised BMP
// when the name ends in .bmp; false when the file cannot be written (like cv::imwrite into a missing directory)
bool imwrite(const std::string &path, const Mat &img);
// 1-pixel outline of `r` (the part inside the image), cv::rectangle(img, r, color, 1, 8, 0) for a grey image
void rectangle(Mat &img, const Rect &r, const Scalar &color, int thickness = 1, int lineType = 8, int shift = 0);

} // namespace cv

#endif // ABUB_USE_OPENCV
#endif
