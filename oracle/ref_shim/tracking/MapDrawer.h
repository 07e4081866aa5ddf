#include "tracking_stubs.h"
