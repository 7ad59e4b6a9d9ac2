// cfr_tail.hpp — host tail of Query (see cfr_tail.cpp)
#pragma once

#include <vector>

#include "cfr_device.hpp"

namespace cfr {

uint64_t tax_lca(const Taxonomy &t, const std::vector<uint64_t> &ids);
void tax_lca_children(const Taxonomy &t, const std::vector<uint64_t> &ids, uint64_t lca, std::vector<uint64_t> &children);
void tax_reduce(const Taxonomy &t, const std::vector<uint64_t> &ids, int k, std::vector<uint64_t> &out,
                std::vector<std::vector<uint64_t>> *children = nullptr);

// --expand-taxid (Classifier.hpp:792-838): spans runs parallel to the match array, ids holds ORIGINAL tax ids
struct ExpandedLists { std::vector<cfr_span> spans; std::vector<uint64_t> ids; };

void classify_read(const HostIndex &h, const cfr_hit *hits, size_t nhits, const uint64_t *row_begin, const uint64_t *row_vals,
                   int32_t query_len, cfr_result &res, std::vector<cfr_match> &matches, ExpandedLists *expanded = nullptr);

void classify_batch_tail(const HostIndex &h, const DeviceIndex::BatchOut &b, size_t n, int threads, cfr_result *results,
                         std::vector<cfr_match> &matches, ExpandedLists *expanded = nullptr);

const char *tax_rank_string(uint8_t rank);

// SDUST pre-step (cfr_dust.cpp): the production form (bounded state) and the literal form of the reference's scan
void dust_mask(uint8_t *s, size_t n);
void dust_mask_literal(uint8_t *s, size_t n);

}  // namespace cfr
