// interrupts.h -- SIGINT levels of rtl_power_fftw (/root/reference/src/interrupts.h:25-35,
// interrupts.cxx:25-60; man page "DESCRIPTION"): the first Ctrl+C lets the current
// frequency scan finish, the second ends the running acquisition as soon as
// possible, the third reaches the default handler and terminates the process.
#ifndef RPF_HOST_INTERRUPTS_H
#define RPF_HOST_INTERRUPTS_H

#include <atomic>

namespace rpf_host {

enum class InterruptState { Neutral = 0, FinishPass = 1, FinishNow = 2 };

extern std::atomic<int> interrupts;
void set_CtrlC_handler(bool install);
// true once at least `level` interrupts arrived; announces each new level once on stderr
bool checkInterrupt(InterruptState level);

}  // namespace rpf_host
#endif
