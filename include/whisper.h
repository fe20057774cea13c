/* whisper.h - the name the reference's callers include (sys/wrapper.h, examples): the MI355X backend's C ABI under it. */
#include "whisper_amd.h"
