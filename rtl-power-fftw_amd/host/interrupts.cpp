#include "interrupts.h"

#include <csignal>
#include <iostream>

namespace rpf_host {

std::atomic<int> interrupts(0);

namespace {
extern "C" void on_sigint(int)
{
    // after the second Ctrl+C the handler steps aside: a third one terminates
    if (interrupts.fetch_add(1) + 1 >= static_cast<int>(InterruptState::FinishNow)) set_CtrlC_handler(false);
}
}  // namespace

void set_CtrlC_handler(bool install)
{
    struct sigaction action;
    action.sa_handler = install ? on_sigint : SIG_DFL;
    sigemptyset(&action.sa_mask);
    action.sa_flags = 0;
    sigaction(SIGINT, &action, nullptr);
}

bool checkInterrupt(InterruptState level)
{
    static int announced = 0;
    const int seen = interrupts.load();
    for (; announced < seen; ++announced) {
        if (announced + 1 == static_cast<int>(InterruptState::FinishPass))
            std::cerr << "Interrupted, will try to finish this pass." << std::endl;
        else if (announced + 1 == static_cast<int>(InterruptState::FinishNow))
            std::cerr << "Interrupted, finishing now." << std::endl;
    }
    return seen >= static_cast<int>(level);
}

}  // namespace rpf_host
