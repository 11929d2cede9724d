// tests/native/agent_match_driver.cpp -- runs csrc/ssd_agent_match.hpp (the rule that maps a HIP device to its HSA agent) over
// fake agent tables.  Compiled with g++ by tests/test_agent_match_cpu.py; prints one line per case, "ok" last.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ssd_agent_match.hpp"

using namespace ssd::aql;

static AgentRecord cpu() { AgentRecord r; r.type = kAgentCpu; return r; }
static AgentRecord gpu(uint32_t domain, uint32_t bus, uint32_t dev, uint32_t fn, const char *uuid, const char *name = "gfx950", uint32_t cus = 256) {
    AgentRecord r;
    r.type = kAgentGpu; r.has_bdf = true; r.domain = domain; r.bdf = (bus << 8) | (dev << 3) | fn;
    if (uuid) std::strncpy(r.uuid, uuid, sizeof r.uuid - 1);
    std::strncpy(r.name, name, sizeof r.name - 1);
    r.cu_count = cus;
    return r;
}
static DeviceRecord device(int ordinal, uint32_t domain, uint32_t bus, uint32_t dev, const char *uuid, bool has_pci = true) {
    DeviceRecord d;
    d.ordinal = ordinal; d.has_pci = has_pci; d.domain = domain; d.bus = bus; d.dev = dev;
    if (uuid) std::strncpy(d.uuid, uuid, sizeof d.uuid - 1);
    std::strncpy(d.arch, "gfx950:sramecc+:xnack-", sizeof d.arch - 1);
    d.cu_count = 256;
    return d;
}
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::printf("FAILED line %d: %s\n", __LINE__, #cond); return 1; } \
    } while (0)

int main() {
    // an 8-GPU node as the HSA runtime lists it: two CPU agents, then the GPUs -- in an order that is NOT HIP's
    const uint32_t buses[8] = {0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xe5, 0xf5};
    const int hsa_order[8] = {3, 0, 1, 2, 7, 6, 5, 4};                      // HSA's i-th GPU is HIP device hsa_order[i]
    char uuids[8][24];
    for (int i = 0; i < 8; ++i) std::snprintf(uuids[i], sizeof uuids[i], "GPU-%016x", 0xabc000u + (unsigned)i);
    std::vector<AgentRecord> node = {cpu(), cpu()};
    for (int i = 0; i < 8; ++i) node.push_back(gpu(0, buses[hsa_order[i]], 0, 0, uuids[hsa_order[i]]));
    for (int n = 0; n < 8; ++n) {                                           // every rank finds ITS device
        const AgentMatch m = match_agent(node, device(n, 0, buses[n], 0, uuids[n]));
        CHECK(m.gpu >= 2 && m.cpu == 0 && std::strcmp(m.by, "PCI address") == 0);
        CHECK((node[m.gpu].bdf >> 8) == buses[n] && hsa_order[m.gpu - 2] == n);
    }
    std::printf("8 GPUs by PCI address: ok\n");
    {   // a PCI function other than 0 (HIP reports none): still the same device; of two functions the lowest
        std::vector<AgentRecord> t = {cpu(), gpu(0, 0x21, 0, 3, nullptr), gpu(0, 0x22, 0, 5, nullptr), gpu(0, 0x22, 0, 2, nullptr)};
        AgentMatch m = match_agent(t, device(0, 0, 0x21, 0, nullptr));
        CHECK(m.gpu == 1 && std::strcmp(m.by, "PCI address") == 0);
        m = match_agent(t, device(1, 0, 0x22, 0, nullptr));
        CHECK(m.gpu == 3);
        // same bus and device in another PCI domain is another device
        m = match_agent(t, device(0, 1, 0x21, 0, nullptr, true));
        CHECK(std::strcmp(m.by, "PCI address") != 0);
    }
    std::printf("PCI function != 0: ok\n");
    {   // no PCI address on either side -> UUID
        std::vector<AgentRecord> t = node;
        for (auto &a : t) a.has_bdf = false;
        for (int n = 0; n < 8; ++n) {
            const AgentMatch m = match_agent(t, device(n, 0, buses[n], 0, uuids[n]));
            CHECK(m.gpu >= 2 && std::strcmp(m.by, "UUID") == 0 && hsa_order[m.gpu - 2] == n);
        }
        // the device's PCI attributes unreadable: UUID as well
        const AgentMatch m = match_agent(node, device(5, 0, 0, 0, uuids[5], /*has_pci=*/false));
        CHECK(m.gpu >= 2 && std::strcmp(m.by, "UUID") == 0 && hsa_order[m.gpu - 2] == 5);
    }
    std::printf("missing BDF -> UUID: ok\n");
    {   // neither PCI address nor UUID: the ordinal -- only unfiltered, and only if architecture and CU count agree
        std::vector<AgentRecord> t = {cpu(), gpu(0, 1, 0, 0, nullptr), gpu(0, 2, 0, 0, nullptr), gpu(0, 3, 0, 0, nullptr, "gfx942", 304)};
        for (auto &a : t) a.has_bdf = false;
        DeviceRecord d = device(1, 0, 9, 0, nullptr, false);
        AgentMatch m = match_agent(t, d);
        CHECK(m.gpu == 2 && std::strcmp(m.by, "ordinal") == 0);
        d.filtered = true;                                                  // HIP_VISIBLE_DEVICES set: HIP's numbering is not HSA's
        m = match_agent(t, d);
        CHECK(m.gpu < 0 && m.why.find("VISIBLE_DEVICES") != std::string::npos);
        d = device(2, 0, 9, 0, nullptr, false);                             // agent number 2 is another architecture
        m = match_agent(t, d);
        CHECK(m.gpu < 0 && m.why.find("gfx942") != std::string::npos);
        std::strncpy(t[3].name, "gfx950", sizeof t[3].name - 1);            // same architecture, another CU count (a partitioned device)
        m = match_agent(t, d);
        CHECK(m.gpu < 0 && m.why.find("compute units") != std::string::npos);
        d = device(7, 0, 9, 0, nullptr, false);                             // more HIP devices than GPU agents
        m = match_agent(t, d);
        CHECK(m.gpu < 0 && m.why.find("no GPU agent number 7") != std::string::npos);
    }
    std::printf("ordinal fallback and its refusals: ok\n");
    {   // no CPU agent: a clean failure string, not a match
        std::vector<AgentRecord> t = {gpu(0, 5, 0, 0, uuids[0])};
        const AgentMatch m = match_agent(t, device(0, 0, 5, 0, uuids[0]));
        CHECK(m.gpu == 0 && m.cpu < 0 && m.why.find("no CPU agent") != std::string::npos);
        const AgentMatch e = match_agent({}, device(0, 0, 5, 0, uuids[0]));
        CHECK(e.gpu < 0 && !e.why.empty());
    }
    std::printf("no CPU agent: ok\n");
    CHECK(arch_agrees("gfx950", "gfx950:sramecc+:xnack-") && arch_agrees("gfx950", "gfx950") && !arch_agrees("gfx95", "gfx950:x") &&
          !arch_agrees("gfx942", "gfx950:sramecc+"));
    std::printf("ok\n");
    return 0;
}
