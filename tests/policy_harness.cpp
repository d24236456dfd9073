// CPU harness around sdf_tools_amd/csrc/sdfgpu_policy.hpp (host-only: no HIP): the test compiles this file with g++ into a
// shared library and drives the dense tier's policy through sequences of builds and (possibly late) reports.
#include <cstdint>

#include "../sdf_tools_amd/csrc/sdfgpu_policy.hpp"

using sdfgpu::DensePlan;
using sdfgpu::DensePolicy;
using sdfgpu::ReportedBuild;

extern "C" {

void* pol_new(int dense_retry) {
    DensePolicy* p = new DensePolicy();
    p->dense_retry = dense_retry;
    return p;
}
void pol_free(void* h) { delete static_cast<DensePolicy*>(h); }

// One build: returns bit 0 dense enqueued, 1 fix-up mode, 2 KD3 in KD's place, 3 staged, 4 the handle trusts its dense tier
// (cheap stand-by behind it).  `take_report` = the previous report has been consumed, so this build's identity is remembered for
// the next one (build_device_impl: `report = ... && !flags_pending`).
int pol_build(void* h, int eligible, int generic, int d3_shape_ok, int vb, int take_report) {
    DensePolicy* p = static_cast<DensePolicy*>(h);
    const DensePlan plan = p->plan(eligible != 0, generic != 0, d3_shape_ok != 0, vb != 0);
    const int out = (plan.dense ? 1 : 0) | (plan.fix ? 2 : 0) | (plan.dense3 ? 4 : 0) | (plan.staged ? 8 : 0) |
                    ((p->expect_dense && plan.dense) ? 16 : 0);
    if (take_report) p->prev = ReportedBuild{plan.dense, generic != 0 && plan.dense, plan.fix_mode_build(), plan.staged};
    return out;
}
// The build just planned takes the report slot: remember what it was (build_device_impl does this when `report` is true).
void pol_remember(void* h, int dense, int generic, int fix_mode, int staged) {
    static_cast<DensePolicy*>(h)->prev = ReportedBuild{dense != 0, generic != 0, fix_mode != 0, staged != 0};
}
// The report of the remembered build arrives.
void pol_report(void* h, int uncertified, int fix_needed, int kd_uncertified) {
    static_cast<DensePolicy*>(h)->consume_report(uncertified != 0, fix_needed != 0, kd_uncertified != 0);
}
int pol_state(void* h, int which) {
    DensePolicy* p = static_cast<DensePolicy*>(h);
    switch (which) {
        case 0: return p->fix_mode; case 1: return p->fix_trust; case 2: return p->dense_skip; case 3: return p->dense_backoff;
        case 4: return p->expect_dense; case 5: return p->fix_clean;
    }
    return -1;
}

// ---- the far-field habit (round 5) ------------------------------------------------------------------------------------------
void* far_new(int mode) {
    sdfgpu::FarHabit* f = new sdfgpu::FarHabit();
    f->set_mode(mode);
    return f;
}
void far_free(void* h) { delete static_cast<sdfgpu::FarHabit*>(h); }
// one build: 1 = the far-field pair without probes
int far_plan(void* h, int selectable, int forced) { return static_cast<sdfgpu::FarHabit*>(h)->plan(selectable != 0, forced != 0) ? 1 : 0; }
void far_report(void* h, int far_y, int far_x) { static_cast<sdfgpu::FarHabit*>(h)->consume_report(far_y != 0, far_x != 0); }
void far_reset(void* h) { static_cast<sdfgpu::FarHabit*>(h)->reset(); }
void far_set_mode(void* h, int mode) { static_cast<sdfgpu::FarHabit*>(h)->set_mode(mode); }
int far_streak(void* h) { return static_cast<sdfgpu::FarHabit*>(h)->streak; }

}  // extern "C"
