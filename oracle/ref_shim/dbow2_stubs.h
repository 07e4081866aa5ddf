// oracle/ref_shim/dbow2_stubs.h -- TEST INFRASTRUCTURE: what the reference's Thirdparty/DBoW2 needs from OpenCV beyond mini_cv.h -- the YAML
// persistence classes its (virtual, hence always instantiated) save / load members name.  The vocabulary of the tests is read with the
// reference's own loadFromTextFile; the cv::FileStorage route aborts when reached.
#ifndef YGZ_ORACLE_REF_SHIM_DBOW2_STUBS_H
#define YGZ_ORACLE_REF_SHIM_DBOW2_STUBS_H
#include "mini_cv.h"
#include <fstream>
#include <sstream>
#include <string>
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
