/* A C99 consumer of include/hello_mi355x.h: what a maintainer binding the library from C (or cgo / JNI / any FFI that reads
 * the header) sees.  Compiled by tests/test_loader_abi.py with -std=c99 -Wall -Wextra -pedantic -Werror and linked against
 * hello_amd/libhello_mi355x.so; prints one line per struct / field -- "name sizeof" or "struct.field offset size" -- which the
 * test compares with the ctypes mirrors of hello_amd/engine.py and hello_amd/records.py, then exercises the entry points that
 * need no GPU: the version, and an engine creation that must be refused with HELLO_ERR_ARG and a message. */
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "hello_mi355x.h"

#define STRUCT(T) printf("struct " #T " %zu\n", sizeof(T))
#define FIELD(T, f) printf("field " #T "." #f " %zu %zu\n", offsetof(T, f), sizeof(((T*)0)->f))

int main(void) {
    hello_engine* engine = (hello_engine*)0x1;
    int rc;

    STRUCT(hello_op);
    FIELD(hello_op, kind); FIELD(hello_op, domain); FIELD(hello_op, src0); FIELD(hello_op, src1); FIELD(hello_op, dst);
    FIELD(hello_op, res); FIELD(hello_op, cin); FIELD(hello_op, cout); FIELD(hello_op, k); FIELD(hello_op, stride);
    FIELD(hello_op, pad); FIELD(hello_op, lin); FIELD(hello_op, lout); FIELD(hello_op, flags); FIELD(hello_op, seg);
    FIELD(hello_op, c1); FIELD(hello_op, a0); FIELD(hello_op, a1); FIELD(hello_op, w_off); FIELD(hello_op, b_off);

    STRUCT(hello_buffer);
    FIELD(hello_buffer, domain); FIELD(hello_buffer, floats_per_row);

    STRUCT(hello_model_desc);
    FIELD(hello_model_desc, abi_version); FIELD(hello_model_desc, window); FIELD(hello_model_desc, channels0);
    FIELD(hello_model_desc, channels1); FIELD(hello_model_desc, n_experts); FIELD(hello_model_desc, has_meta);
    FIELD(hello_model_desc, uses_ref); FIELD(hello_model_desc, n_buffers); FIELD(hello_model_desc, buffers);
    FIELD(hello_model_desc, n_ops); FIELD(hello_model_desc, ops);

    STRUCT(hello_site_table);
    FIELD(hello_site_table, n_sites); FIELD(hello_site_table, alleles_per_site); FIELD(hello_site_table, allele_text);
    FIELD(hello_site_table, allele_text_off); FIELD(hello_site_table, n_chromosomes); FIELD(hello_site_table, chromosome_text);
    FIELD(hello_site_table, chromosome_text_off); FIELD(hello_site_table, chromosome_of_site); FIELD(hello_site_table, start);
    FIELD(hello_site_table, stop); FIELD(hello_site_table, ref_windows); FIELD(hello_site_table, ref_window_off);
    FIELD(hello_site_table, window_start); FIELD(hello_site_table, genome); FIELD(hello_site_table, genome_len);
    FIELD(hello_site_table, keep);

    STRUCT(hello_features_format);
    FIELD(hello_features_format, meta_prefix); FIELD(hello_features_format, meta_prefix_len);
    FIELD(hello_features_format, meta_suffix); FIELD(hello_features_format, meta_suffix_len);

    STRUCT(hello_records_view);
    FIELD(hello_records_view, n_sites); FIELD(hello_records_view, n_shards); FIELD(hello_records_view, shard_vcf);
    FIELD(hello_records_view, shard_vcf_off); FIELD(hello_records_view, mean_vcf); FIELD(hello_records_view, mean_vcf_off);
    FIELD(hello_records_view, mean_position); FIELD(hello_records_view, features); FIELD(hello_records_view, features_off);
    FIELD(hello_records_view, n_records); FIELD(hello_records_view, best_pair); FIELD(hello_records_view, best_p);
    FIELD(hello_records_view, qual);

    STRUCT(hello_site_slot_layout);
    FIELD(hello_site_slot_layout, header); FIELD(hello_site_slot_layout, rpa0); FIELD(hello_site_slot_layout, rpa1);
    FIELD(hello_site_slot_layout, ref); FIELD(hello_site_slot_layout, logits); FIELD(hello_site_slot_layout, meta);
    FIELD(hello_site_slot_layout, post); FIELD(hello_site_slot_layout, err); FIELD(hello_site_slot_layout, reads);
    FIELD(hello_site_slot_layout, read_capacity);

    STRUCT(hello_site_server_config);
    FIELD(hello_site_server_config, window); FIELD(hello_site_server_config, channels0); FIELD(hello_site_server_config, channels1);
    FIELD(hello_site_server_config, n_experts); FIELD(hello_site_server_config, has_meta); FIELD(hello_site_server_config, uses_ref);
    FIELD(hello_site_server_config, max_clients); FIELD(hello_site_server_config, max_batch_sites);
    FIELD(hello_site_server_config, group_launches); FIELD(hello_site_server_config, reserved);
    FIELD(hello_site_server_config, slot_bytes); FIELD(hello_site_server_config, idle_exit_s); FIELD(hello_site_server_config, linger_s);
    FIELD(hello_site_server_config, info_json);

    STRUCT(hello_site_server_stats);
    FIELD(hello_site_server_stats, launches); FIELD(hello_site_server_stats, sites); FIELD(hello_site_server_stats, errors);
    FIELD(hello_site_server_stats, largest_launch); FIELD(hello_site_server_stats, clients_seen);

    printf("const HELLO_SITE_PROTOCOL %d\nconst HELLO_SITE_MAX_ALLELES %d\n", HELLO_SITE_PROTOCOL, HELLO_SITE_MAX_ALLELES);
    {
        hello_site_slot_layout lay;
        hello_site_server* server = (hello_site_server*)0x1;
        rc = hello_site_slot_layout_of(150, 6, 7, 1 << 20, &lay);
        printf("call hello_site_slot_layout_of %d %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld\n", rc, (long long)lay.header, (long long)lay.rpa0,
               (long long)lay.rpa1, (long long)lay.ref, (long long)lay.logits, (long long)lay.meta, (long long)lay.post, (long long)lay.err,
               (long long)lay.reads, (long long)lay.read_capacity);
        if (rc != HELLO_OK) return 5;
        if (hello_site_slot_layout_of(150, 6, 0, 512, &lay) != HELLO_ERR_ARG) return 6;       /* a slot that cannot hold one read */
        rc = hello_site_server_create(NULL, NULL, NULL, &server);
        printf("call hello_site_server_create(NULL) %d message %s\n", rc, hello_last_error());
        if (rc != HELLO_ERR_ARG) return 7;
        hello_site_server_stop(NULL);           /* no-ops by contract */
        hello_site_server_destroy(NULL);
    }

    printf("const HELLO_ABI_VERSION %d\n", HELLO_ABI_VERSION);
    printf("const HELLO_IN_DEVICE %d\nconst HELLO_OUT_DEVICE %d\nconst HELLO_LAYOUT_RCL %d\n", HELLO_IN_DEVICE, HELLO_OUT_DEVICE,
           HELLO_LAYOUT_RCL);
    printf("const HELLO_OP_READCONV_FUSED %d\nconst HELLO_OP_XATTN_FRONT %d\nconst HELLO_BUF_FIRST_SCRATCH %d\n",
           (int)HELLO_OP_READCONV_FUSED, (int)HELLO_OP_XATTN_FRONT, HELLO_BUF_FIRST_SCRATCH);

    printf("call hello_abi_version %d\n", hello_abi_version());
    rc = hello_engine_create(NULL, NULL, 0, 0, &engine);
    printf("call hello_engine_create(NULL) %d engine_reset %d message %s\n", rc, engine == NULL, hello_last_error());
    if (rc != HELLO_ERR_ARG || engine != NULL || strlen(hello_last_error()) == 0) return 2;
    rc = hello_engine_create(NULL, NULL, 0, 0, NULL);
    printf("call hello_engine_create(out=NULL) %d message %s\n", rc, hello_last_error());
    if (rc != HELLO_ERR_ARG) return 3;
    hello_engine_destroy(NULL);                 /* a no-op by contract */
    hello_records_destroy(NULL);
    if (hello_abi_version() != HELLO_ABI_VERSION) return 4;
    return 0;
}
