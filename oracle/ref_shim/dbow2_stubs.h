// oracle/ref_shim/dbow2_stubs.h -- TEST INFRASTRUCTURE: what the reference's Thirdparty/DBoW2 needs from OpenCV beyond mini_cv.h -- the YAML
// persistence classes its (virtual, hence always instantiated) save / load members name.  The vocabulary of the tests is read with the
// reference's own loadFromTextFile; the cv::FileStorage route aborts when reached.
#ifndef YGZ_ORACLE_REF_SHIM_DBOW2_STUBS_H
#define YGZ_ORACLE_REF_SHIM_DBOW2_STUBS_H
#include "mini_cv.h"
#include <fstream>
#include <sstream>
#include <string>
#ifdef YGZ_REF_TRACKING
#include <map>
namespace cv {
// a WORKING minimal reader for the boundary build that compiles src/Tracking.cc: its constructor reads the camera / extractor settings through
// cv::FileStorage (src/Tracking.cc:83-213).  "Key: value" lines of a flat YAML file; a missing key reads as 0 / "" as OpenCV's empty node does.
struct FileNode {
    std::string v;
    FileNode() {}
    explicit FileNode(const std::string &s) : v(s) {}
    FileNode operator[](const char *) const { mini_cv_unsupported("cv::FileNode[]"); }
    FileNode operator[](const std::string &) const { mini_cv_unsupported("cv::FileNode[]"); }
    FileNode operator[](int) const { mini_cv_unsupported("cv::FileNode[]"); }
    size_t size() const { return v.empty() ? 0 : 1; }
    bool empty() const { return v.empty(); }
    operator int() const { return v.empty() ? 0 : (int) std::strtod(v.c_str(), nullptr); }
    operator float() const { return v.empty() ? 0.f : (float) std::strtod(v.c_str(), nullptr); }
    operator double() const { return v.empty() ? 0.0 : std::strtod(v.c_str(), nullptr); }
    operator std::string() const { return v; }
};
struct FileStorage {
    enum { READ = 0, WRITE = 1 };
    std::map<std::string, std::string> kv;
    bool ok = false;
    FileStorage(const std::string &path, int) {
        std::ifstream f(path.c_str());
        ok = (bool) f;
        std::string line;
        while (std::getline(f, line)) {
            const size_t c = line.find(':');
            if (c == std::string::npos || line[0] == '%' || line[0] == '#') continue;
            std::string k = line.substr(0, c), val = line.substr(c + 1);
            while (!val.empty() && (val[0] == ' ' || val[0] == '\t')) val.erase(0, 1);
            while (!val.empty() && (val.back() == ' ' || val.back() == '\r')) val.pop_back();
            kv[k] = val;
        }
    }
    bool isOpened() const { return ok; }
    FileNode operator[](const char *k) const { auto it = kv.find(k); return it == kv.end() ? FileNode() : FileNode(it->second); }
    FileNode operator[](const std::string &k) const { return (*this)[k.c_str()]; }
    void release() {}
};
template <class T> inline FileStorage &operator<<(FileStorage &, const T &) { mini_cv_unsupported("cv::FileStorage <<"); }
enum { CV_RGB2GRAY_ = 7 };
}  // namespace cv
#define CV_RGB2GRAY 7
#define CV_BGR2GRAY 6
#define CV_RGBA2GRAY 11
#define CV_BGRA2GRAY 10
namespace cv {
inline void cvtColor(const Mat &, Mat &, int) { mini_cv_unsupported("cv::cvtColor (the boundary test feeds gray images)"); }
}
#else
namespace cv {
struct FileNode {
    FileNode operator[](const char *) const { mini_cv_unsupported("cv::FileNode"); }
    FileNode operator[](const std::string &) const { mini_cv_unsupported("cv::FileNode"); }
    FileNode operator[](int) const { mini_cv_unsupported("cv::FileNode"); }
    size_t size() const { mini_cv_unsupported("cv::FileNode"); }
    operator int() const { mini_cv_unsupported("cv::FileNode"); }
    operator double() const { mini_cv_unsupported("cv::FileNode"); }
    operator std::string() const { mini_cv_unsupported("cv::FileNode"); }
};
struct FileStorage {
    enum { READ = 0, WRITE = 1 };
    FileStorage(const char *, int) {}
    bool isOpened() const { return false; }
    FileNode operator[](const char *) const { mini_cv_unsupported("cv::FileStorage"); }
    FileNode operator[](const std::string &) const { mini_cv_unsupported("cv::FileStorage"); }
};
template <class T> inline FileStorage &operator<<(FileStorage &, const T &) { mini_cv_unsupported("cv::FileStorage <<"); }
}  // namespace cv
#endif
#endif
