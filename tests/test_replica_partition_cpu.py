"""CPU: partition arithmetic of the REPLICA layout behind the C ABI (vdb_replica_query_block, no GPU call): the blocks of all
ranks tile [0, nq) in rank order, each at most ceil(nq / world) long, and equal shard.replica_query_slice -- the torch host's
rule (HNSW = replicas, contiguous query blocks, one fixed-size all-gather)."""
import pytest


@pytest.mark.parametrize("nq", [0, 1, 2, 7, 8, 9, 1000, 8192, 100003])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_replica_blocks_tile_the_queries(nq, world):
    from lab_1806_vec_db_amd.shard import replica_query_slice
    from lab_1806_vec_db_amd.sharded import replica_query_block

    per = -(-nq // world)
    pos = 0
    for r in range(world):
        q0, q1 = replica_query_block(nq, world, r)
        assert (q0, q1) == replica_query_slice(nq, world, r)
        assert q0 == min(nq, pos) and q0 <= q1 <= nq and q1 - q0 <= per
        pos = q1
    assert pos == nq


def test_replica_block_rejects_bad_rank():
    import lab_1806_vec_db_amd as vdb
    from lab_1806_vec_db_amd.sharded import replica_query_block

    with pytest.raises(vdb.VdbError):
        replica_query_block(10, 2, 2)
    with pytest.raises(vdb.VdbError):
        replica_query_block(10, 0, 0)
