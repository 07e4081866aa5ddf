// stand-in for <opencv2/core/core.hpp>: see ../../mini_cv.h
#include "../../mini_cv.h"
