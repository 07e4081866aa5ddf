// stand-in for <opencv2/features2d/features2d.hpp>: see ../../mini_cv.h
#include "../../mini_cv.h"
