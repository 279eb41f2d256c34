"""TEST INFRASTRUCTURE ONLY: CPU oracle for the MAPF environment hot path (see mapf_oracle.c).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only; the product
package (mapf_rl_amd) must never import this."""
