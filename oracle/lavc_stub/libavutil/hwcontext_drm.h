#include "ug_lavc_stub.h"
