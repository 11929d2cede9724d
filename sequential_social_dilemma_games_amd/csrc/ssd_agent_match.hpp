// csrc/ssd_agent_match.hpp -- which HSA agent is HIP device n?  A pure function over plain records (no HIP, no HSA): the
// library's own dispatch path (ssd_aql.hip, device_ctx) fills the records from the two runtimes and acts on the verdict; a CPU
// test (tests/test_agent_match_cpu.py) compiles this header with g++ and runs it over fake tables -- eight GPUs, PCI functions
// other than 0, a missing PCI address, filtered device lists, no CPU agent -- because on the one-GPU boxes this code is built
// on, `device = local_rank != 0` never runs (run_scripts/train_moa.py:127-128 is the reference's data-parallel axis; bench.py
// --gpus 8 is its first execution here).
//
// The rule, in order:
//   1. PCI address: same domain, bus and device; the function number is ignored (HIP reports no function; an HSA BDFID is
//      bus << 8 | device << 3 | function), the lowest matching function wins.
//   2. UUID: the agent whose "GPU-<16 hex digits>" string HIP reports for the device.
//   3. Ordinal: the n-th GPU agent -- only while nothing filters or reorders the devices HIP shows (HIP_VISIBLE_DEVICES /
//      CUDA_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES unset), and only if what both runtimes say about the device agrees: the
//      architecture name and the number of compute units.  A wrong match would send dispatches to another GPU.
// A match also needs a CPU agent (the kernarg memory pool lives there).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace ssd {
namespace aql {

enum AgentType { kAgentCpu = 0, kAgentGpu = 1, kAgentOther = 2 };

struct AgentRecord {            // one HSA agent, in hsa_iterate_agents order
    int type = kAgentOther;
    bool has_bdf = false;       // HSA_AMD_AGENT_INFO_BDFID / _DOMAIN could be read
    uint32_t bdf = 0, domain = 0;
    char uuid[24] = {};         // HSA_AMD_AGENT_INFO_UUID, "GPU-<16 hex digits>"; empty: none
    char name[64] = {};         // HSA_AGENT_INFO_NAME, e.g. "gfx950"
    uint32_t cu_count = 0;      // HSA_AMD_AGENT_INFO_COMPUTE_UNIT_COUNT; 0: unknown
};

struct DeviceRecord {           // what HIP says about the device to match
    int ordinal = 0;            // the HIP device number
    bool has_pci = false;       // hipDeviceAttributePci{DomainID,BusId,DeviceId} could be read
    uint32_t domain = 0, bus = 0, dev = 0;
    char uuid[24] = {};         // "GPU-" + hipDeviceGetUuid's 16 bytes; empty: none
    char arch[64] = {};         // hipDeviceProp_t::gcnArchName, e.g. "gfx950:sramecc+:xnack-"
    uint32_t cu_count = 0;      // hipDeviceProp_t::multiProcessorCount; 0: unknown
    bool filtered = false;      // a *_VISIBLE_DEVICES variable is set: HIP's numbering is not HSA's
};

struct AgentMatch {
    int gpu = -1, cpu = -1;     // indices into the agent list; gpu < 0: no match
    const char *by = "";        // "PCI address" | "UUID" | "ordinal"
    std::string why;            // when gpu < 0 (or cpu < 0): what failed
};

inline bool arch_agrees(const char *hsa_name, const char *hip_arch) {
    // "gfx950" against "gfx950:sramecc+:xnack-": the HSA name must be the HIP name up to the first ':'
    if (!hsa_name[0] || !hip_arch[0]) return true;                 // (nothing to compare: no objection)
    const char *colon = std::strchr(hip_arch, ':');
    const size_t n = colon ? (size_t)(colon - hip_arch) : std::strlen(hip_arch);
    return std::strlen(hsa_name) == n && std::strncmp(hsa_name, hip_arch, n) == 0;
}

inline AgentMatch match_agent(const std::vector<AgentRecord> &agents, const DeviceRecord &d) {
    AgentMatch m;
    for (size_t i = 0; i < agents.size(); ++i)
        if (agents[i].type == kAgentCpu) { m.cpu = (int)i; break; }
    // 1. PCI address (function ignored, lowest function first)
    if (d.has_pci) {
        uint32_t best_fn = 8;
        for (size_t i = 0; i < agents.size(); ++i) {
            const AgentRecord &a = agents[i];
            if (a.type != kAgentGpu || !a.has_bdf || a.domain != d.domain) continue;
            if ((a.bdf >> 8) != d.bus || ((a.bdf >> 3) & 31u) != d.dev) continue;
            const uint32_t fn = a.bdf & 7u;
            if (fn < best_fn) { best_fn = fn; m.gpu = (int)i; }
        }
        if (m.gpu >= 0) m.by = "PCI address";
    }
    // 2. UUID
    if (m.gpu < 0 && d.uuid[0]) {
        for (size_t i = 0; i < agents.size(); ++i)
            if (agents[i].type == kAgentGpu && agents[i].uuid[0] && std::strncmp(agents[i].uuid, d.uuid, sizeof d.uuid) == 0) {
                m.gpu = (int)i; m.by = "UUID";
                break;
            }
    }
    // 3. ordinal, cross-checked
    std::string ordinal_note;
    if (m.gpu < 0) {
        if (d.filtered) ordinal_note = "; ordinal not tried: a *_VISIBLE_DEVICES variable filters the devices";
        else {
            int seen = 0, idx = -1;
            for (size_t i = 0; i < agents.size(); ++i)
                if (agents[i].type == kAgentGpu) { if (seen == d.ordinal) { idx = (int)i; break; } ++seen; }
            if (idx < 0) ordinal_note = "; there is no GPU agent number " + std::to_string(d.ordinal);
            else if (!arch_agrees(agents[idx].name, d.arch))
                ordinal_note = std::string("; GPU agent number ") + std::to_string(d.ordinal) + " is a " + agents[idx].name + ", the device a " + d.arch;
            else if (agents[idx].cu_count && d.cu_count && agents[idx].cu_count != d.cu_count)
                ordinal_note = "; GPU agent number " + std::to_string(d.ordinal) + " has " + std::to_string(agents[idx].cu_count) +
                               " compute units, the device " + std::to_string(d.cu_count);
            else { m.gpu = idx; m.by = "ordinal"; }
        }
    }
    if (m.gpu < 0) {
        char pci[48] = "unknown";
        if (d.has_pci) std::snprintf(pci, sizeof pci, "%04x:%02x:%02x", d.domain, d.bus, d.dev);
        m.why = "no HSA agent matches HIP device " + std::to_string(d.ordinal) + " (PCI address " + pci + ", UUID " +
                (d.uuid[0] ? std::string(d.uuid, strnlen(d.uuid, sizeof d.uuid)) : std::string("none")) + ordinal_note + ")";
        m.by = "";
    } else if (m.cpu < 0) {
        m.why = "the HSA runtime lists no CPU agent (no kernarg memory pool)";
    }
    return m;
}

}  // namespace aql
}  // namespace ssd
