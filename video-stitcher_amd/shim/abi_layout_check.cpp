// Checks ms_image against the reference's OWN definition of PtrStepSz<T> (sources/modules/core/include/opencv2/core/cuda_types.hpp:95-120 --
// header-only, includes no generated file): same offsets for data / step / cols / rows, so a kernel argument block written for PtrStepSz<T>
// is the prefix of an ms_image.  Built and run by __graft_entry__.build() and tests/test_abi.py where /root/reference exists (this container);
// the full cv::cuda::GpuMat declaration (core/cuda.hpp) cannot be included: core.hpp -> base.hpp needs the cmake-generated opencv_modules.hpp.
#include <cstddef>
#include <cstdio>
#include "opencv2/core/cuda_types.hpp"
#include "../../include/ms_stitch.h"

int main()
{
    unsigned char buf[16];
    cv::cuda::PtrStepSz<unsigned char> p(7, 9, buf, 48);            // (rows, cols, data, step)
    const char *base = reinterpret_cast<const char *>(&p);
    const long o_data = reinterpret_cast<const char *>(&p.data) - base, o_step = reinterpret_cast<const char *>(&p.step) - base;
    const long o_cols = reinterpret_cast<const char *>(&p.cols) - base, o_rows = reinterpret_cast<const char *>(&p.rows) - base;
    int bad = 0;
    bad += o_data != (long)offsetof(ms_image, data);
    bad += o_step != (long)offsetof(ms_image, step);
    bad += o_cols != (long)offsetof(ms_image, cols);
    bad += o_rows != (long)offsetof(ms_image, rows);
    bad += sizeof(p.step) != sizeof(((ms_image *)0)->step) || sizeof(p.cols) != sizeof(int);
    bad += sizeof(p) > offsetof(ms_image, type) + sizeof(int) || offsetof(ms_image, type) < (size_t)o_rows + sizeof(int);
    ms_image m;
    __builtin_memcpy(&m, &p, sizeof(p));                               // the reinterpretation the header promises
    bad += m.data != buf || m.step != 48 || m.cols != 9 || m.rows != 7;
    std::printf("PtrStepSz<uchar>: data@%ld step@%ld cols@%ld rows@%ld size %zu; ms_image: data@%zu step@%zu cols@%zu rows@%zu type@%zu -> %s\n",
                o_data, o_step, o_cols, o_rows, sizeof(p), offsetof(ms_image, data), offsetof(ms_image, step), offsetof(ms_image, cols),
                offsetof(ms_image, rows), offsetof(ms_image, type), bad ? "MISMATCH" : "layout ok");
    return bad ? 1 : 0;
}
