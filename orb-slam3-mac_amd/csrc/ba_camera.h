// ba_camera.h -- GeometricCamera::project / projectJac on the device, shared by the local BA (ba_kernels.hip) and the inertial local
// BA (iba_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
// GeometricCamera::project / projectJac of camera (fx,fy,cx,cy,model,k): Pinhole.cpp:41-47,81-91; KannalaBrandt8.cpp:52-69,166-195
__device__ __forceinline__ void cam_project(double fx, double fy, double cx, double cy, int model, const double *k, const double *P, double *uv)
{
    if (model == 1) {
        const double x2y2 = P[0] * P[0] + P[1] * P[1];
        const double theta = (double)(float)atan2((double)sqrtf((float)x2y2), (double)(float)P[2]);
        const double psi = (double)(float)atan2((double)(float)P[1], (double)(float)P[0]);
        const double t2 = theta * theta, t3 = theta * t2, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
        const double r = theta + k[0] * t3 + k[1] * t5 + k[2] * t7 + k[3] * t9;
        uv[0] = fx * r * cos(psi) + cx; uv[1] = fy * r * sin(psi) + cy;
    } else { uv[0] = fx * P[0] / P[2] + cx; uv[1] = fy * P[1] / P[2] + cy; }
}
__device__ __forceinline__ void cam_project_jac(double fx, double fy, int model, const double *k, const double *P, double *J)
{
    const double x = P[0], y = P[1], z = P[2];
    if (model == 1) {
        const double x2 = x * x, y2 = y * y, z2 = z * z, r2 = x2 + y2, r = sqrt(r2), r3 = r2 * r;
        const double theta = atan2(r, z);
        const double t2 = theta * theta, t3 = t2 * theta, t4 = t2 * t2, t5 = t4 * theta, t6 = t2 * t4, t7 = t6 * theta, t8 = t4 * t4, t9 = t8 * theta;
        const double f = theta + t3 * k[0] + t5 * k[1] + t7 * k[2] + t9 * k[3];
        const double fd = 1 + 3 * k[0] * t2 + 5 * k[1] * t4 + 7 * k[2] * t6 + 9 * k[3] * t8;
        J[0] = fx * (fd * z * x2 / (r2 * (r2 + z2)) + f * y2 / r3);
        J[3] = fy * (fd * z * y * x / (r2 * (r2 + z2)) - f * y * x / r3);
        J[1] = fx * (fd * z * y * x / (r2 * (r2 + z2)) - f * y * x / r3);
        J[4] = fy * (fd * z * y2 / (r2 * (r2 + z2)) + f * x2 / r3);
        J[2] = -fx * fd * x / (r2 + z2); J[5] = -fy * fd * y / (r2 + z2);
    } else {
        J[0] = fx / z; J[1] = 0; J[2] = -fx * x / (z * z);
        J[3] = 0; J[4] = fy / z; J[5] = -fy * y / (z * z);
    }
}
