// stand-in for <opencv2/imgproc/imgproc.hpp>: see ../../mini_cv.h
#include "../../mini_cv.h"
