/* DeviceGuard.h -- internal to the host classes (Stixels, RoadEstimation).
 * Runs a public method on the device of the object's buffers and puts the caller's current device
 * back on every exit path (SetDevice() contract in Stixels.hpp / RoadEstimation.h): a stream, a
 * copy and a synchronisation all belong to the device that is current when they are issued. */
#ifndef INSTANCESTIXELS_AMD_DEVICEGUARD_H_
#define INSTANCESTIXELS_AMD_DEVICEGUARD_H_

#include "InstanceStixels/util.h"

namespace {
class DeviceGuard {
public:
    explicit DeviceGuard(int device) {
        if (device < 0) return;
        IS_CHECK_RETURN(is_get_device(&m_prev));
        if (m_prev != device) {
            IS_CHECK_RETURN(is_set_device(device));
            m_switched = true;
        }
    }
    ~DeviceGuard() {
        if (m_switched) (void)is_set_device(m_prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;

private:
    int m_prev = -1;
    bool m_switched = false;
};
}  // namespace

#endif
